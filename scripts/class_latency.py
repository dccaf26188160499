#!/usr/bin/env python3
"""Per-query latency of the drop-in class API, the way scripts/query.py uses it (one query per call):
HybridIndexer.search(query, top_k=fetch_k) + reranker.rerank(query, results, top_k), on a synthetic text corpus.

    python scripts/class_latency.py [--chunks 100000] [--queries 30] [--reranker crossencoder|simple|none]
"""
import argparse
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'domain-rag-system_amd'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--chunks', type=int, default=100_000)
    ap.add_argument('--queries', type=int, default=30)
    ap.add_argument('--reranker', default='crossencoder', choices=['crossencoder', 'simple', 'none'])
    ap.add_argument('--rerank-top-n', type=int, default=50)
    ap.add_argument('--top-k', type=int, default=5)
    args = ap.parse_args()
    import torch
    from indexing import BM25Indexer, HybridIndexer, MedicalEmbedder, VectorIndexer
    from reranker import build_reranker
    rng = np.random.default_rng(0)
    words = ['w%05d' % i for i in range(20000)]
    p = 1.0 / np.arange(1, len(words) + 1) ** 1.07
    p /= p.sum()
    lens = np.clip(np.rint(rng.lognormal(np.log(45), 0.6, args.chunks)), 2, 128).astype(int)
    picks = rng.choice(len(words), size=int(lens.sum()), p=p)
    chunks, at = [], 0
    for i, n in enumerate(lens.tolist()):
        chunks.append({'chunk_id': 'c%d' % i, 'doc_id': 'd%d' % (i // 4), 'source': 'synthetic', 'chunk_type': 'text',
                       'text': ' '.join(words[j] for j in picks[at:at + n]), 'metadata': {}})
        at += n
    emb = MedicalEmbedder(model_name='random:bert-base', device='cuda')
    vi = VectorIndexer('bench', emb, use_in_memory=True)
    hy = HybridIndexer(vi, BM25Indexer(), storage_path='/tmp/class_latency_idx')
    t0 = time.perf_counter()
    hy.index_chunks(chunks)
    torch.cuda.synchronize()
    print(f'index build: {args.chunks} chunks in {time.perf_counter() - t0:.1f} s', flush=True)
    rr = None
    if args.reranker != 'none':
        rr = build_reranker(args.reranker, top_n=args.rerank_top_n, cross_model='random:minilm-l6', embedder=emb,
                            **({'vector_indexer': vi} if args.reranker == 'simple' else {}))
    fetch_k = max(args.top_k, args.rerank_top_n if rr is not None else args.top_k)       # scripts/query.py:139
    times = {'search': [], 'rerank': [], 'total': []}
    for qi in range(args.queries + 3):
        q = ' '.join(words[j] for j in rng.choice(len(words), size=int(rng.integers(4, 13)), p=p))
        t0 = time.perf_counter()
        res = hy.search(q, top_k=fetch_k, fusion_method='rrf')
        t1 = time.perf_counter()
        if rr is not None:
            res = rr.rerank(q, res, top_k=args.top_k)
        t2 = time.perf_counter()
        if qi >= 3:
            times['search'].append((t1 - t0) * 1e3)
            times['rerank'].append((t2 - t1) * 1e3)
            times['total'].append((t2 - t0) * 1e3)
        assert len(res) == args.top_k
    print(f'{args.reranker}: per query, median of {args.queries}: search {statistics.median(times["search"]):.2f} ms, '
          f'rerank {statistics.median(times["rerank"]):.2f} ms, total {statistics.median(times["total"]):.2f} ms '
          f'(max {max(times["total"]):.2f} ms)', flush=True)


if __name__ == '__main__':
    main()
