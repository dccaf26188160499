#!/usr/bin/env python3
"""Query the hybrid index, with optional rerank.

Same flags and behaviour as the reference's ``scripts/query.py``: the index is re-attached
from ``<indices_dir>/index_metadata.json`` + ``bm25_index.pkl``; retrieval runs on the
preprocessed query with ``fetch_k = max(top_k, rerank_top_n)`` when a reranker is chosen;
reranking scores the ORIGINAL query text; ``--output`` writes
``{query, preprocessed, num_results, results}``.
"""
import argparse
import json
import sys
from pathlib import Path

sys.path.append(str(Path(__file__).resolve().parent.parent / 'domain-rag-system_amd'))

from indexing import MedicalEmbedder, VectorIndexer, BM25Indexer, HybridIndexer  # noqa: E402
from preprocessing import MedicalTermNormalizer, QueryPreprocessor  # noqa: E402
from reranker import build_reranker  # noqa: E402


def load_index_metadata(metadata_path: str):
    with open(metadata_path, 'r', encoding='utf-8') as f:
        return json.load(f)


def load_hybrid_indexer(indices_dir: str, metadata: dict, drug_mapping_path=None) -> HybridIndexer:
    """Attach to the stores written by build_index.py."""
    indices = Path(indices_dir)
    embedder = MedicalEmbedder(model_name=metadata['embedding_model'], batch_size=32)
    vector_indexer = VectorIndexer(collection_name=metadata['collection_name'], embedder=embedder,
                                   storage_path=metadata.get('vector_storage_path', 'data/vector_db'),
                                   distance_metric=metadata['distance_metric'])
    normalizer = MedicalTermNormalizer()
    if drug_mapping_path and Path(drug_mapping_path).exists():
        with open(drug_mapping_path, 'r', encoding='utf-8') as f:
            normalizer.drug_mapping = json.load(f)
    params = metadata.get('bm25_params', {'k1': 1.5, 'b': 0.75})
    bm25_indexer = BM25Indexer(k1=params.get('k1', 1.5), b=params.get('b', 0.75), medical_normalizer=normalizer)
    bm25_path = indices / 'bm25_index.pkl'
    if not bm25_path.exists():
        raise FileNotFoundError(f"BM25 index not found: {bm25_path}")
    bm25_indexer.load(str(bm25_path))
    return HybridIndexer(vector_indexer=vector_indexer, bm25_indexer=bm25_indexer, storage_path=indices_dir)


def main(args):
    metadata_path = Path(args.indices_dir) / 'index_metadata.json'
    if not metadata_path.exists():
        print(f"Error: Index metadata not found at {metadata_path}")
        print("Please run build_index.py first to build the index.")
        return None
    metadata = load_index_metadata(str(metadata_path))
    hybrid = load_hybrid_indexer(args.indices_dir, metadata, args.drug_mapping_path)

    normalizer = MedicalTermNormalizer()
    if args.drug_mapping_path and Path(args.drug_mapping_path).exists():
        with open(args.drug_mapping_path, 'r') as f:
            normalizer.drug_mapping = json.load(f)
    preprocessed = QueryPreprocessor(medical_normalizer=normalizer).preprocess(args.query)
    search_query = preprocessed.get('cleaned') or args.query
    if preprocessed.get('normalized') and len(preprocessed['normalized']) < len(search_query) * 2:
        search_query = preprocessed['normalized']

    fetch_k = max(args.top_k, args.rerank_top_n if args.reranker_kind != 'none' else args.top_k)
    results = hybrid.search(query=search_query, top_k=fetch_k, fusion_method=args.fusion_method,
                            vector_weight=args.vector_weight, bm25_weight=args.bm25_weight, filters=args.filters)
    if args.reranker_kind != 'none' and results:
        rr = build_reranker(kind=args.reranker_kind, top_n=args.rerank_top_n, cross_model=args.cross_model,
                            embedder=hybrid.vector_indexer.embedder, vector_indexer=hybrid.vector_indexer)
        if rr is not None:
            results = rr.rerank(args.query, results, top_k=args.top_k)

    for i, r in enumerate(results[:args.top_k], 1):
        extra = f" rerank={r['rerank_score']:.4f}" if 'rerank_score' in r else ''
        print(f"{i}. {r['chunk_id']} score={r['score']:.6f}{extra} | {(r.get('text') or '')[:100]}")
    output = {'query': args.query, 'preprocessed': preprocessed, 'num_results': min(args.top_k, len(results)),
              'results': results[:args.top_k]}
    if args.output:
        with open(args.output, 'w', encoding='utf-8') as f:
            json.dump(output, f, indent=2, ensure_ascii=False)
        print(f"Results saved to {args.output}")
    return output


def build_parser():
    p = argparse.ArgumentParser(description='Query the medical RAG index')
    p.add_argument('--query', type=str, required=True)
    p.add_argument('--indices_dir', type=str, default='data/indices')
    p.add_argument('--drug_mapping_path', type=str, default='data/processed_for_our_rag/drug_mapping.json')
    p.add_argument('--top_k', type=int, default=5)
    p.add_argument('--fusion_method', type=str, default='rrf', choices=['rrf', 'weighted'])
    p.add_argument('--vector_weight', type=float, default=0.5)
    p.add_argument('--bm25_weight', type=float, default=0.5)
    p.add_argument('--filters', type=str, default=None, help='metadata filters as a JSON object')
    p.add_argument('--reranker_kind', type=str, default='none', choices=['none', 'simple', 'crossencoder'])
    p.add_argument('--rerank_top_n', type=int, default=50)
    p.add_argument('--cross_model', type=str, default='cross-encoder/ms-marco-MiniLM-L-6-v2')
    p.add_argument('--output', type=str, default=None)
    return p


if __name__ == '__main__':
    a = build_parser().parse_args()
    a.filters = json.loads(a.filters) if a.filters else None
    main(a)
