"""GPU: the screen-then-rescore rerank (BatchSearcher(screen_encoder=...)) returns what the exact cross-encoder alone
returns - same final rows in the same order, rerank scores from the exact forward - while scoring only a fraction of
the pairs exactly (in one stage with a two-sided margin, or in two with a bound from the first stage's exact logits);
the self-check widens a margin that was too small.  The reference scores every pair in float32 and
sorts (src/reranker/rerank.py:29-43,79-84): only the order and the top_k scores are observable, and those must not move."""
import numpy as np
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu


def _world(dev, init, n=20000, vocab=3000):
    import rag_hip
    from indexing import bert
    from indexing.bm25_indexer import build_arrays
    rag_hip.init(0)
    corpus = synth.dense_corpus(n, 128, seed=71)
    term, doc, ln = synth.sparse_corpus(n, vocab, seed=72, mean_len=45)
    arr = build_arrays(term, doc, ln, vocab, 1.5, 0.75)
    dense = rag_hip.DenseIndex(corpus)
    bm25 = rag_hip.Bm25Index(arr.indptr, arr.rows, arr.impacts(), arr.idf, n)
    exact, _, _ = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.float32, native=True, split_gemm=True,
                                  split_dtype=torch.float16, init=init)
    cheap, _, _ = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.float16, native=True, init=init)
    rng = np.random.default_rng(73)
    off = np.concatenate([[0], np.cumsum(ln + 2)]).astype(np.int64)
    tok = rng.integers(1000, 30522, size=int(off[-1])).astype(np.int32)
    return dict(dense=dense, bm25=bm25, exact=exact, cheap=cheap, tok=torch.from_numpy(tok).to(dev),
                off=torch.from_numpy(off).to(dev), term=term, doc=doc, ln=ln, vocab=vocab, dim=128)


def _queries(w, b, seed, dev):
    import rag_hip
    rng = np.random.default_rng(seed)
    qlen = rng.integers(8, 25, size=b)
    ce_tok = [rng.integers(1000, 30522, size=int(n)).astype(np.int32) for n in qlen]
    ce_off = np.concatenate([[0], np.cumsum(qlen)]).astype(np.int64)
    qterms, qoff = rag_hip.Bm25Index.flatten(synth.sparse_queries(w['term'], w['doc'], w['ln'], w['vocab'], b, seed=seed + 1))
    t = lambda a: torch.from_numpy(a).to(dev)
    return t(qterms), t(qoff), int(qoff[-1]), t(np.concatenate(ce_tok)), t(ce_off)


class _Vectors:
    """Stands in for the bi-encoder: seeded unit vectors, so the test needs no BERT-base forward."""

    def __init__(self, b, dim, seed, dev):
        g = torch.Generator(device=dev).manual_seed(seed)
        self.v = torch.nn.functional.normalize(torch.randn((b, dim), generator=g, device=dev), dim=1)


@pytest.mark.parametrize('init,two_stage', [('hf-default', True), ('trained-scale', True), ('hf-default', False)])
def test_screened_rerank_equals_the_exact_rerank(init, two_stage):
    from indexing.batch import BatchSearcher, TokenStore
    import rag_hip
    dev = torch.device('cuda:0')
    w = _world(dev, init)
    b, top_n = 256, 30
    mk = lambda screen: BatchSearcher(w['dense'], w['bm25'], cross_encoder=w['exact'], screen_encoder=screen,
                                      chunk_tokens=TokenStore(w['tok'], w['off'], 0), top_k=5, rerank_top_n=top_n,
                                      screen_two_stage=two_stage)
    plain, screened = mk(None), mk(w['cheap'])
    assert screened.screen_encoder is not None and plain.screen_encoder is None
    fractions = []
    for it in range(3):
        qterms, qoff, n_terms, ce_tok, ce_off = _queries(w, b, 500 + 10 * it, dev)
        qv = _Vectors(b, w['dim'], 900 + it, dev).v
        k_list = 2 * top_n
        vec = plain.dense_search(qv, k_list)
        bm = plain.bm25_search(qterms, qoff, n_terms, k_list)
        fused = plain.fuse(vec, bm, top_n)
        want = plain.finish(fused, ce_tok, ce_off)
        before = dict(screened.screen_stats)
        got = screened.finish(fused, ce_tok, ce_off)
        torch.cuda.synchronize()
        mask = screened.last_rescored
        assert mask is not None and plain.last_rescored is None
        # exact logits wherever a pair was rescored (a pair's logit does not depend on the batch it is computed in)
        assert float((got['logits'] - want['logits'])[mask].abs().max()) < 2e-6
        # final lists: same rows in the same order, unless two exact logits deciding the order are within that noise
        srt = torch.sort(want['logits'].double(), dim=1, descending=True).values
        gap = (srt[:, :5] - srt[:, 1:6]).min(dim=1).values
        same = (got['rows'] == want['rows']).all(dim=1)
        assert bool((same | (gap < 4e-6)).all()), int((~same).sum())
        assert torch.equal(got['count'], want['count'])
        assert float((got['rerank_score'] - want['rerank_score'])[same].abs().max()) < 2e-6
        # the top k by exact logit were all rescored
        top = torch.topk(want['logits'], 5, dim=1).indices
        assert bool(torch.gather(mask, 1, top).all())
        st = screened.screen_stats
        fractions.append((st['rescored'] - before['rescored']) / (st['pairs'] - before['pairs']))
    assert fractions[0] == 1.0                       # the first batch has no error estimate yet: every pair is rescored
    assert max(fractions[1:]) < 0.75                 # later ones rescore a fraction (measured: 0.2-0.5 at 30 pairs per query)
    rag_hip.check_faults()
    w['dense'].close()
    w['bm25'].close()


def test_a_margin_that_was_too_small_is_widened():
    from indexing.batch import BatchSearcher, TokenStore
    dev = torch.device('cuda:0')
    w = _world(dev, 'hf-default')
    b, top_n = 200, 30
    mk = lambda screen: BatchSearcher(w['dense'], w['bm25'], cross_encoder=w['exact'], screen_encoder=screen,
                                      chunk_tokens=TokenStore(w['tok'], w['off'], 0), top_k=5, rerank_top_n=top_n)
    plain, screened = mk(None), mk(w['cheap'])
    qterms, qoff, n_terms, ce_tok, ce_off = _queries(w, b, 77, dev)
    qv = _Vectors(b, w['dim'], 78, dev).v
    fused = plain.fuse(plain.dense_search(qv, 60), plain.bm25_search(qterms, qoff, n_terms, 60), top_n)
    want = plain.finish(fused, ce_tok, ce_off)
    screened._screen_err = 1e-9                       # an estimate far below the screen's real error
    got = screened.finish(fused, ce_tok, ce_off)
    torch.cuda.synchronize()
    assert screened.screen_stats['widened'] >= 1
    assert screened._screen_err > 1e-6               # replaced by what the batch measured
    srt = torch.sort(want['logits'].double(), dim=1, descending=True).values
    gap = (srt[:, :5] - srt[:, 1:6]).min(dim=1).values
    same = (got['rows'] == want['rows']).all(dim=1)
    assert bool((same | (gap < 4e-6)).all())
    w['dense'].close()
    w['bm25'].close()


def test_small_batches_take_the_exact_path_only():
    from indexing.batch import BatchSearcher, TokenStore
    dev = torch.device('cuda:0')
    w = _world(dev, 'hf-default', n=5000)
    s = BatchSearcher(w['dense'], w['bm25'], cross_encoder=w['exact'], screen_encoder=w['cheap'],
                      chunk_tokens=TokenStore(w['tok'], w['off'], 0), top_k=5, rerank_top_n=20)
    qterms, qoff, n_terms, ce_tok, ce_off = _queries(w, 8, 5, dev)
    qv = _Vectors(8, w['dim'], 6, dev).v
    fused = s.fuse(s.dense_search(qv, 40), s.bm25_search(qterms, qoff, n_terms, 40), 20)
    s.finish(fused, ce_tok, ce_off)
    assert s.last_rescored is None and s.screen_stats['batches'] == 0
    w['dense'].close()
    w['bm25'].close()


def test_screened_rerank_with_ragged_and_empty_pools():
    """Pools shorter than rerank_top_n, shorter than top_k, and empty (fused rows -1 behind the count): the screen must
    rescore whole pools that cannot fill a top k, never choose an empty slot, and agree with the exact rerank."""
    from indexing.batch import BatchSearcher, TokenStore
    import rag_hip
    dev = torch.device('cuda:0')
    w = _world(dev, 'hf-default', n=8000)
    b, t = 320, 24
    rng = np.random.default_rng(91)
    cnt = rng.integers(0, t + 1, size=b)
    cnt[:6] = [0, 1, 4, 5, 6, t]
    rows = np.full((b, t), -1, np.int64)
    for i in range(b):
        rows[i, :cnt[i]] = rng.choice(8000, size=cnt[i], replace=False)
    fused_row = torch.from_numpy(rows).to(dev)
    fused_cnt = torch.from_numpy(cnt.astype(np.int32)).to(dev)
    _, _, _, ce_tok, ce_off = _queries(w, b, 33, dev)
    mk = lambda screen: BatchSearcher(w['dense'], w['bm25'], cross_encoder=w['exact'], screen_encoder=screen,
                                      chunk_tokens=TokenStore(w['tok'], w['off'], 0), top_k=5, rerank_top_n=t)
    plain, screened = mk(None), mk(w['cheap'])
    for it in range(2):                                   # the second call is the screened one
        iw, cw, lw = plain.rerank(ce_tok, ce_off, fused_row, fused_cnt)
        ig, cg, lg = screened.rerank(ce_tok, ce_off, fused_row, fused_cnt)
        torch.cuda.synchronize()
        assert torch.equal(cg, cw) and torch.equal(cg.cpu(), torch.clamp(fused_cnt.cpu(), max=5))
        live = torch.arange(t, device=dev)[None, :] < fused_cnt[:, None]
        srt = torch.sort(torch.where(live, lw, torch.full_like(lw, -1e30)).double(), dim=1, descending=True).values
        gap = (srt[:, :5] - srt[:, 1:6]).abs().min(dim=1).values
        same = (ig == iw).all(dim=1)
        assert bool((same | (gap < 4e-6)).all())
        mask = screened.last_rescored
        assert bool((mask <= live).all())                 # no empty slot was scored as a pair
        short = fused_cnt <= 5
        assert bool((mask[short] == live[short]).all())   # pools that cannot fill a top 5 are rescored whole
    assert screened.screen_stats['rescored'] < screened.screen_stats['pairs']
    rag_hip.check_faults()
    w['dense'].close()
    w['bm25'].close()


def test_a_screen_that_overflows_decides_nothing():
    """Non-finite cheap logits (a float16 overflow inside the screening forward): every pair is rescored and the result
    is the exact rerank's."""
    from indexing.batch import BatchSearcher, TokenStore
    dev = torch.device('cuda:0')
    w = _world(dev, 'hf-default', n=8000)
    b, top_n = 160, 30
    mk = lambda screen: BatchSearcher(w['dense'], w['bm25'], cross_encoder=w['exact'], screen_encoder=screen,
                                      chunk_tokens=TokenStore(w['tok'], w['off'], 0), top_k=5, rerank_top_n=top_n)
    plain, screened = mk(None), mk(w['cheap'])
    qterms, qoff, n_terms, ce_tok, ce_off = _queries(w, b, 41, dev)
    qv = _Vectors(b, w['dim'], 42, dev).v
    fused = plain.fuse(plain.dense_search(qv, 60), plain.bm25_search(qterms, qoff, n_terms, 60), top_n)
    want = plain.finish(fused, ce_tok, ce_off)
    screened.finish(fused, ce_tok, ce_off)                   # first batch: sets the error estimate
    inner = screened._score_pairs

    def spoiled(model, *a, **kw):
        out = inner(model, *a, **kw)
        if model is screened.screen_encoder:
            out = out.clone()
            out[::7] = float('inf')
            out[3::11] = float('nan')
        return out
    screened._score_pairs = spoiled
    before = dict(screened.screen_stats)
    got = screened.finish(fused, ce_tok, ce_off)
    torch.cuda.synchronize()
    st = screened.screen_stats
    assert st['rescored'] - before['rescored'] == st['pairs'] - before['pairs']       # everything was rescored
    assert torch.equal(got['rows'], want['rows'])
    assert float((got['rerank_score'] - want['rerank_score']).abs().max()) < 2e-6
    assert np.isfinite(screened._screen_err)
    w['dense'].close()
    w['bm25'].close()


def test_tile_order_and_mini_batch_cuts_do_not_change_a_logit(monkeypatch):
    """The float16 forward runs its pairs in the order that fills 256-token tiles and cuts the stream into mini-batches
    at tile boundaries: the same pairs through mini-batches of a few tiles each, through one mini-batch, and through
    the two-kernel route (no tiles) give the same logits to float16 rounding."""
    from indexing.batch import BatchSearcher, TokenStore
    dev = torch.device('cuda:0')
    w = _world(dev, 'hf-default', n=6000)
    b, t = 96, 30
    rng = np.random.default_rng(17)
    cnt = rng.integers(0, t + 1, size=b)
    rows = np.full((b, t), -1, np.int64)
    for i in range(b):
        rows[i, :cnt[i]] = rng.choice(6000, size=cnt[i], replace=False)
    fused_row = torch.from_numpy(rows).to(dev)
    _, _, _, ce_tok, ce_off = _queries(w, b, 19, dev)
    s = BatchSearcher(w['dense'], w['bm25'], cross_encoder=w['cheap'], chunk_tokens=TokenStore(w['tok'], w['off'], 0),
                      top_k=5, rerank_top_n=t)
    pq = torch.arange(b, dtype=torch.int32, device=dev).repeat_interleave(t)
    pr = fused_row.reshape(-1)

    def logits():
        with torch.no_grad():
            out = s._score_pairs(w['cheap'], ce_tok, ce_off, pq, pr)
        torch.cuda.synchronize()
        return out
    assert s._tiles_wanted(w['cheap'])
    one = logits()
    monkeypatch.setenv('RAG_CE_TOKENS', '700')               # two or three tiles per mini-batch
    many = logits()
    monkeypatch.delenv('RAG_CE_TOKENS')
    live = (pr >= 0)
    # (not bit for bit: the library GEMMs of the pooler and the classifier see another row count per mini-batch)
    assert float((one[live] - many[live]).abs().max()) < 5e-4
    w['cheap'].bert.fused_qkv_attention = False              # projection and attention as two kernels, pairs longest first
    try:
        two = logits()
    finally:
        w['cheap'].bert.fused_qkv_attention = True
    assert float((one[live] - two[live]).abs().max()) < 2e-3
    import rag_hip
    rag_hip.check_faults()
    w['dense'].close()
    w['bm25'].close()
