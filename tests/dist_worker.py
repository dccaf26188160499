"""Worker for the world_size-2 gloo tests (TEST INFRASTRUCTURE): one rank of the row-sharded
batched query path.  On the CPU the GPU entry points are replaced by the oracle-backed doubles; with
device 'cuda' the real kernels run, every rank on GPU 0 (the exchange still goes over gloo)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'domain-rag-system_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def make_world(n=600, dim=64, vocab=300, seed=0):
    from tests import synth
    corpus = synth.dense_corpus(n, dim, seed=seed)
    term, doc, ln = synth.sparse_corpus(n, vocab, seed=seed + 1, mean_len=12)
    term, _, v = synth.remap_first_appearance(term, vocab)
    rng = np.random.default_rng(seed + 2)
    canon = np.arange(n, dtype=np.uint32)
    for d in (40, 300, 301, 555):
        canon[d] = canon[int(rng.integers(0, d))]
    clen = ln + 2
    off = np.concatenate([[0], np.cumsum(clen)]).astype(np.int64)
    tok = rng.integers(1000, 4096, size=int(off[-1])).astype(np.int32)
    b = 9
    q_ids = np.zeros((b, 8), np.int64)
    q_ids[:, 0], q_ids[:, 7] = 101, 102
    q_ids[:, 1:7] = rng.integers(1000, 4096, size=(b, 6))
    bm_q = synth.sparse_queries(term, doc, ln, v, b, seed=seed + 3)
    return dict(corpus=corpus, term=term, doc=doc, ln=ln, v=v, canon=canon, tok=tok, off=off, q_ids=q_ids, bm_q=bm_q)


def run_rank(rank, world, port, out_path, fusion='rrf', metric='cosine', reranker='cross', device='cpu'):
    import types
    from tests import fake_backend
    import rag_hip
    for name in () if device != 'cpu' else ('DenseIndex', 'Bm25Index', 'fuse_rrf', 'fuse_weighted', 'fuse_rrf_dev', 'fuse_weighted_dev',
                 'topk_segmented', 'topk_segmented_dev', 'topk_stable_f64', 'merge_topk_dev', 'ce_pair_len_dev',
                 'ce_pack_dev', 'check_faults'):
        setattr(rag_hip, name, getattr(fake_backend, name))
    from indexing import bert
    from indexing.batch import BatchSearcher, TokenStore
    from indexing.bm25_indexer import build_arrays
    dist = None
    force = os.environ.get('RAG_DIST_FORCE') == '1'      # single rank over RCCL: every exchange runs, each an identity
    if world > 1 or force:
        import torch.distributed as dist
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        if force and world == 1 and device != 'cpu':
            torch.cuda.set_device(0)
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
    w = make_world()
    n = w['corpus'].shape[0]
    lo, hi = n * rank // world, n * (rank + 1) // world
    sel = (w['doc'] >= lo) & (w['doc'] < hi)
    stats = None
    if world > 1 or force:
        df = np.bincount(np.unique(w['term'] * n + w['doc']) // n, minlength=w['v'])
        stats = dict(df=df, total_len=int(w['ln'].sum()), n_docs=n)
    arr = build_arrays(w['term'][sel], w['doc'][sel] - lo, w['ln'][lo:hi], w['v'], 1.5, 0.75, global_stats=stats)
    dense = rag_hip.DenseIndex(w['corpus'][lo:hi], metric=metric, row_base=lo)
    bm25 = rag_hip.Bm25Index(arr.indptr, arr.rows, arr.impacts(), arr.idf, hi - lo, row_base=lo)
    bi, _, _ = bert.load_model('random:tiny', 'bi', device)
    ce, _, _ = bert.load_model('random:tiny', 'cross', device)
    native_screen = None
    if 'native' in reranker:
        # the packed route of the real kernels: float32-equivalent MiniLM-L6 forward, float16 screening forward
        ce, _, _ = bert.load_model('random:minilm-l6', 'cross', device, dtype=torch.float32, native=True, split_gemm=True,
                                   split_dtype=torch.float16)
        if reranker.endswith('screened'):
            native_screen, _, _ = bert.load_model('random:minilm-l6', 'cross', device, dtype=torch.float16, native=True)
    if device != 'cpu':
        rag_hip.init(0)
    dv = lambda a: (a if torch.is_tensor(a) else torch.from_numpy(a)).to(device)
    screen = native_screen
    if reranker == 'cross-screened':
        # the screen-then-rescore rerank: the cheap model is the cross-encoder with its weights rounded to float16
        import copy
        screen = copy.deepcopy(ce)
        with torch.no_grad():
            for p in screen.parameters():
                p.copy_(p.half().float())
    s = BatchSearcher(dense, bm25, canon=dv(w['canon']), bi_encoder=bi, screen_encoder=screen, screen_min_pairs=1,
                      cross_encoder=ce if reranker.startswith('cross') else None, simple_rerank=reranker == 'simple',
                      chunk_tokens=TokenStore(dv(w['tok']), dv(w['off'])),
                      top_k=4, rerank_top_n=11, fusion_method=fusion, vector_weight=0.6, bm25_weight=0.4,
                      ce_pairs_per_batch=16, dist=dist)
    qterms, qoff = rag_hip.Bm25Index.flatten(w['bm_q'])
    ids = dv(w['q_ids'])
    ce_tok = dv(w['q_ids'][:, 1:7].astype(np.int32).reshape(-1).copy())
    ce_off = dv(torch.arange(ids.shape[0] + 1, dtype=torch.int64) * 6)
    qterms, qoff = dv(qterms), dv(qoff)
    s.split_embed = False                  # first every rank embeds every query ...
    out = s.search(ids, torch.ones_like(ids), qterms, qoff, int(qoff[-1]),
                   ce_tok, ce_off)
    if world > 1 or force:
        # ... then the default: each rank embeds a slice and the slices are exchanged; same answer
        s.split_embed = True
        again = s.search(ids, torch.ones_like(ids), qterms, qoff, int(qoff[-1]),
                         ce_tok, ce_off)
        for k, v in out.items():
            if k == 'logits' and screen is not None:
                continue                   # (pairs the screen left out keep their cheap logit there, by design)
            if torch.is_tensor(v):
                if v.dtype.is_floating_point:
                    assert torch.allclose(v, again[k], atol=1e-5), k
                else:
                    assert torch.equal(v, again[k]), k
    if screen is not None:
        # a third search: the first had no error estimate (every pair rescored), later ones screen for real
        third = s.search(ids, torch.ones_like(ids), qterms, qoff, int(qoff[-1]), ce_tok, ce_off)
        assert s.screen_stats['batches'] >= 2 and s.last_rescored is not None
        assert torch.equal(third['rows'], out['rows']), 'the screened rerank changed the final rows'
        out = dict(third, rescored_fraction=torch.tensor(s.screen_stats['rescored'] / max(s.screen_stats['pairs'], 1)))
    if rank == 0:
        torch.save({k: v.cpu() for k, v in out.items() if torch.is_tensor(v)}, out_path)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    run_rank(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5],
             sys.argv[6] if len(sys.argv) > 6 else 'cosine', sys.argv[7] if len(sys.argv) > 7 else 'cross', sys.argv[8] if len(sys.argv) > 8 else 'cpu')
