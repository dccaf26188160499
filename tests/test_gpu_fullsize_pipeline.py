"""GPU: the composed pipelines of the two 1M x 768 single-GPU configurations at full size, through
BatchSearcher (the batched form of scripts/query.py:139-162), on the synthetic shard bench.py times.

  * hybrid vector + BM25 with RRF, top_k = 5, no reranker: size-independent properties of every
    query's result plus the oracle on a sample of queries;
  * hybrid + MiniLM-L6 cross-encoder, rerank_top_n = 50, B = 1024: the default (float32, split-fp16
    products) cross-encoder path against a plain PyTorch float32 forward on the same inputs — final
    top-5 ids in order; and the bfloat16 path beside it, which does not hold that order.
"""
import argparse

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ARGS = dict(rows=1_000_000, dim=768, vocab=200_000, batch=1024, top_k=5)


@pytest.fixture(scope='module')
def shard():
    import bench
    import rag_hip
    from indexing import bert
    rag_hip.init(0)
    dev = torch.device('cuda:0')
    args = argparse.Namespace(**ARGS)
    rows, dense, bm25, tok, off, canon, host = bench.build_shard(args, 0, 1, dev, None)
    bi, _, _ = bert.load_model('random:bert-base', 'bi', dev, native=True, split_gemm=True)
    queries = bench.make_queries(args, 1, dev)[0]
    yield dict(args=args, rows=rows, dense=dense, bm25=bm25, tok=tok, off=off, canon=canon, host=host, bi=bi,
               q=queries, dev=dev)
    dense.close()
    bm25.close()


def _searcher(sh, ce, top_n, screen=None):
    from indexing.batch import BatchSearcher, TokenStore
    return BatchSearcher(sh['dense'], sh['bm25'], canon=sh['canon'], bi_encoder=sh['bi'], cross_encoder=ce, screen_encoder=screen,
                         chunk_tokens=TokenStore(sh['tok'], sh['off'], 0), top_k=5, rerank_top_n=top_n)


def _run(srch, q):
    out = srch.search(q['ids'], q['mask'], q['qterms'], q['qoff'], q['n_terms'], q['ce_tok'], q['ce_off'])
    torch.cuda.synchronize()
    return out


def test_hybrid_rrf_top5_at_one_million_rows(shard):
    """1M x 768, hybrid + RRF, top_k 5 (per-list K = 10).  Every query: fused scores are sums of
    1/(60 + rank) over the ranks its canonical id holds in the two lists, best first, ties in first-appearance
    order.  A sample of queries: lists and fused result identical to the oracle's."""
    from oracle.bm25_ref import BM25OkapiArrays, rank_rows_ref
    from oracle.dense_ref import dense_topk_ref
    from oracle.fusion_ref import rrf_keys
    sh = shard
    q = sh['q']
    s = _searcher(sh, None, 50)
    assert s.fetch_k == 5
    out = _run(s, q)
    qvec = s.embed(q['ids'], q['mask'])
    vs, vr, vc = s.dense_search(qvec, 10)
    bs, br, bc = s.bm25_search(q['qterms'], q['qoff'], q['n_terms'], 10)
    torch.cuda.synchronize()
    canon = sh['host']['canon']
    vr_h, br_h, vc_h, bc_h = vr.cpu().numpy(), br.cpu().numpy(), vc.cpu().numpy(), bc.cpu().numpy()
    vs_h, bs_h = vs.cpu().numpy(), bs.cpu().numpy()
    rows_o, score_o, cnt_o = out['rows'].cpu().numpy(), out['scores'].cpu().numpy(), out['count'].cpu().numpy()
    assert int(vc_h.min()) == 10 and int(bc_h.min()) == 10        # 1M rows: both lists are always full
    for i in range(rows_o.shape[0]):
        assert np.all(np.diff(vs_h[i]) <= 0) and np.all(np.diff(bs_h[i]) <= 0)          # lists best first
        want = rrf_keys([int(canon[r]) for r in vr_h[i]], [int(canon[r]) for r in br_h[i]], top_k=5)
        both = list(vr_h[i]) + list(br_h[i])
        assert rows_o[i, :cnt_o[i]].tolist() == [int(both[p]) for p, _ in want], i
        assert score_o[i, :cnt_o[i]].tolist() == [sc for _, sc in want], i              # fp64, bit for bit
    # the oracle from scratch on a few queries (dense: ids equal wherever fp32 scores are 2e-6 apart)
    rows_h = sh['rows'].cpu().numpy()
    ref = BM25OkapiArrays.from_token_ids(sh['host']['term'], sh['host']['doc'], sh['host']['ln'], sh['args'].vocab)
    qv = qvec.cpu().numpy()
    hq = q['host']
    for i in (0, 1, 511, 1023):
        d_rows, d_scores = dense_topk_ref(rows_h, qv[i], 11)
        if np.min(-np.diff(d_scores)) > 2e-6:
            assert vr_h[i].tolist() == list(d_rows[:10])
        assert np.abs(vs_h[i] - d_scores[:10]).max() < 1e-4
        terms = hq['terms'][hq['qoff'][i]:hq['qoff'][i + 1]].tolist()
        sc = ref.get_scores_ids(terms)
        assert br_h[i].tolist() == list(rank_rows_ref(sc, 10))
        assert np.array_equal(bs_h[i], sc[br_h[i]])


def test_hybrid_cross_encoder_top5_equals_the_fp32_forward_at_config_size(shard):
    """1M x 768, hybrid + RRF + MiniLM-L6 cross-encoder, rerank_top_n 50, 1024 queries: the default path
    (float32 weights, GEMMs and attention as split-fp16 products) must return the top-5 ids, in order, of a
    plain PyTorch float32 cross-encoder on the same fused candidates.  A query is excused only when two of the
    float32 logits deciding its top-5 are closer than eps (summation-order round-off); the count is reported."""
    from indexing import bert
    sh = shard
    dev = sh['dev']
    eps = 2e-5
    ce_fast, _, _ = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.float32, native=True, split_gemm=True,
                                   split_dtype=torch.float16)
    ce_ref, _, _ = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.float32)
    s_fast, s_ref = _searcher(sh, ce_fast, 50), _searcher(sh, ce_ref, 50)
    assert s_fast.ce_packed and not s_ref.ce_packed
    fast, ref = _run(s_fast, sh['q']), _run(s_ref, sh['q'])
    assert torch.equal(fast['fused_row'], ref['fused_row']) and torch.equal(fast['fused_score'], ref['fused_score'])
    assert int(fast['fused_count'].min()) == 50
    assert float((fast['logits'] - ref['logits']).abs().max()) < 5e-6
    srt = torch.sort(ref['logits'].double(), dim=1, descending=True).values
    gap = (srt[:, :5] - srt[:, 1:6]).min(dim=1).values
    same = (fast['rows'] == ref['rows']).all(dim=1)
    excused = int((~same & (gap < eps)).sum())
    unexcused = int((~same & (gap >= eps)).sum())
    print(f'config 4: {int(same.sum())}/1024 top-5 lists identical, {excused} differ with a deciding fp32 gap < {eps}, '
          f'{unexcused} unexcused; {int((gap >= eps).sum())} queries have every deciding gap >= eps')
    assert unexcused == 0
    assert int((gap >= eps).sum()) >= 512            # most queries are decided above round-off: the check is not vacuous
    assert excused <= 64
    # rerank scores of the returned entries are the float32 logits
    assert float((fast['rerank_score'][same] - ref['rerank_score'][same]).abs().max()) < 5e-6

    # the default of the timed path: every pair screened by a float16 forward, the pairs near each query's top 5 rescored
    # by the float32-equivalent forward.  Same bar, same eps: 0 unexcused; scores of the returned entries exact.
    # (two batches: the first has no error estimate and rescores every pair, the second is the screened one)
    ce_h, _, _ = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.float16, native=True)
    s_scr = _searcher(sh, ce_fast, 50, screen=ce_h)
    _run(s_scr, sh['q'])
    assert s_scr.last_rescored is not None and bool(s_scr.last_rescored.all())
    scr = _run(s_scr, sh['q'])
    mask = s_scr.last_rescored
    frac = float(mask.float().mean())
    same_s = (scr['rows'] == ref['rows']).all(dim=1)
    unexcused_s = int((~same_s & (gap >= eps)).sum())
    print(f'config 4, screened: {int(same_s.sum())}/1024 top-5 lists identical, {unexcused_s} unexcused, '
          f'{frac:.3f} of the pairs rescored, within-query screen error {s_scr._screen_err:.2e}')
    assert unexcused_s == 0 and int((~same_s).sum()) <= 64
    assert float((scr['logits'] - ref['logits'])[mask].abs().max()) < 5e-6
    assert float((scr['rerank_score'][same_s] - ref['rerank_score'][same_s]).abs().max()) < 5e-6
    assert bool(torch.gather(mask, 1, torch.topk(ref['logits'], 5, dim=1).indices).all())      # the true top 5 were rescored
    assert frac < 0.5
    assert torch.equal(scr['rows'], fast['rows'])    # and the very lists of the unscreened path

    # the bfloat16 kernels on the same inputs: logits good to bf16 precision only, so the order among
    # candidates whose float32 logits differ by less than that is not held (reported, not asserted equal)
    ce_bf, _, _ = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.bfloat16, native=True)
    bf = _run(_searcher(sh, ce_bf, 50), sh['q'])
    err = float((bf['logits'] - ref['logits']).abs().max())
    same_bf = (bf['rows'] == ref['rows']).all(dim=1)
    print(f'config 4, bfloat16 cross-encoder: max logit error {err:.2e}, {int(same_bf.sum())}/1024 top-5 lists identical')
    assert err < 2e-2
    decided_bf = gap >= 4 * err
    assert bool(same_bf[decided_bf].all())           # where the fp32 gaps exceed the bf16 error the order holds
