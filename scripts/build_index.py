#!/usr/bin/env python3
"""Build the hybrid index (dense store + BM25) from a chunks JSONL file.

Same flags, defaults and output files as the reference's ``scripts/build_index.py``
(``--chunks_path --drug_mapping_path --embedding_model --batch_size --collection_name
--vector_storage_path --distance_metric --bm25_k1 --bm25_b --output_dir``): writes
``<output_dir>/bm25_index.pkl``, ``<output_dir>/index_metadata.json`` and the vector store
under ``--vector_storage_path``.  ``--embedding_model`` also accepts a local checkpoint
directory or ``random:<arch>``; ``index_metadata.json`` additionally records
``vector_storage_path`` so ``query.py`` finds a non-default store.
"""
import argparse
import json
import sys
from pathlib import Path

sys.path.append(str(Path(__file__).resolve().parent.parent / 'domain-rag-system_amd'))

from indexing import MedicalEmbedder, VectorIndexer, BM25Indexer, HybridIndexer  # noqa: E402
from preprocessing import MedicalTermNormalizer  # noqa: E402


def load_chunks(chunks_path: str):
    """One chunk dict per line."""
    with open(chunks_path, 'r', encoding='utf-8') as f:
        return [json.loads(line) for line in f if line.strip()]


def main(args):
    chunks = load_chunks(args.chunks_path)
    print(f"Loaded {len(chunks)} chunks from {args.chunks_path}")
    embedder = MedicalEmbedder(model_name=args.embedding_model, batch_size=args.batch_size)
    vector_indexer = VectorIndexer(collection_name=args.collection_name, embedder=embedder,
                                   storage_path=args.vector_storage_path, distance_metric=args.distance_metric)
    normalizer = MedicalTermNormalizer()
    if Path(args.drug_mapping_path).exists():
        with open(args.drug_mapping_path, 'r', encoding='utf-8') as f:
            normalizer.drug_mapping = json.load(f)
    bm25_indexer = BM25Indexer(k1=args.bm25_k1, b=args.bm25_b, medical_normalizer=normalizer)
    hybrid = HybridIndexer(vector_indexer=vector_indexer, bm25_indexer=bm25_indexer, storage_path=args.output_dir)
    hybrid.index_chunks(chunks)

    out = Path(args.output_dir)
    bm25_path = out / 'bm25_index.pkl'
    hybrid.save_bm25(str(bm25_path))
    metadata = {
        'num_chunks': len(chunks),
        'embedding_model': args.embedding_model,
        'embedding_dim': embedder.embedding_dim,
        'collection_name': args.collection_name,
        'bm25_params': {'k1': args.bm25_k1, 'b': args.bm25_b},
        'distance_metric': args.distance_metric,
        'vector_storage_path': args.vector_storage_path,
    }
    with open(out / 'index_metadata.json', 'w') as f:
        json.dump(metadata, f, indent=2)
    print("Index statistics:")
    print(json.dumps(hybrid.get_stats(), indent=2))
    print(f"- Vector DB: {args.vector_storage_path}\n- BM25 index: {bm25_path}\n- Metadata: {out / 'index_metadata.json'}")


def build_parser():
    p = argparse.ArgumentParser(description='Build hybrid index for medical RAG')
    p.add_argument('--chunks_path', type=str, default='data/processed_for_our_rag/chunks.jsonl')
    p.add_argument('--drug_mapping_path', type=str, default='data/processed_for_our_rag/drug_mapping.json')
    p.add_argument('--embedding_model', type=str, default='pritamdeka/S-PubMedBert-MS-MARCO',
                   help='hub id (local cache only), checkpoint directory, or random:<arch>')
    p.add_argument('--batch_size', type=int, default=32)
    p.add_argument('--collection_name', type=str, default='medical_rag')
    p.add_argument('--vector_storage_path', type=str, default='data/vector_db')
    p.add_argument('--distance_metric', type=str, default='cosine', choices=['cosine', 'euclidean', 'dot'])
    p.add_argument('--bm25_k1', type=float, default=1.5)
    p.add_argument('--bm25_b', type=float, default=0.75)
    p.add_argument('--output_dir', type=str, default='data/indices')
    return p


if __name__ == '__main__':
    main(build_parser().parse_args())
