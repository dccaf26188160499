"""GPU parity: exact dense kNN (librag_hip via the C ABI) against the CPU oracle."""
import numpy as np
import pytest

from oracle.dense_ref import dense_scores_ref, dense_topk_ref, normalize_rows_ref
from tests import synth

pytestmark = pytest.mark.gpu

TOL = 1e-4      # cosine tolerance stated by the north star (fp32)


def _check(corpus, queries, k, metric='cosine', mask=None, dups=False):
    """Both forms of the scan: split-bf16 products (the default where dim % 32 == 0) and fp32 FMA chains."""
    import rag_hip
    idx = rag_hip.DenseIndex(corpus, metric=metric)
    assert idx.split == (corpus.shape[1] % 32 == 0)
    _check_index(idx, corpus, queries, k, metric, mask, dups)
    if idx.split:
        idx.set_split(False)
        _check_index(idx, corpus, queries, k, metric, mask, dups)
    idx.close()


def _check_index(idx, corpus, queries, k, metric, mask, dups):
    score, row, cnt = idx.search(queries, k, row_mask=mask)
    n_ok = corpus.shape[0] if mask is None else int(mask.sum())
    for b in range(queries.shape[0]):
        ref_rows, ref_scores = dense_topk_ref(corpus, queries[b], k, metric=metric, row_mask=mask)
        c = int(cnt[b])
        assert c == min(k, n_ok)
        assert np.all(row[b, c:] == -1)
        got_rows, got_scores = row[b, :c], score[b, :c]
        # scores: within tolerance of the fp32 CPU dot products, position by position
        np.testing.assert_allclose(got_scores, ref_scores, atol=TOL, rtol=0)
        # the list is ordered by the kernel's own scores: best first, ties by row descending
        s = got_scores if metric != 'euclidean' else -got_scores
        assert np.all((s[:-1] > s[1:]) | ((s[:-1] == s[1:]) & (got_rows[:-1] > got_rows[1:])))
        if np.array_equal(got_rows, ref_rows):
            continue
        # ids may differ from the oracle's only where two scores are closer than fp32
        # accumulation-order noise: every such row must itself score within tolerance
        all_s = dense_scores_ref(corpus, queries[b], metric)
        np.testing.assert_allclose(all_s[got_rows], got_scores, atol=TOL, rtol=0)
        gap = np.abs(all_s[got_rows] - ref_scores)
        assert np.all(gap[got_rows != ref_rows] < 2e-6), (got_rows, ref_rows)
        if dups is False:                # one swapped pair at most in a short list, 5 % in a long one
            assert np.sum(got_rows != ref_rows) <= max(2, 0.05 * len(ref_rows))


@pytest.mark.parametrize('n,dim,b,k', [
    (1000, 768, 1, 10), (1000, 768, 7, 10), (5000, 768, 33, 10), (5000, 768, 64, 100),
    (20000, 768, 100, 10), (20000, 768, 200, 100), (3000, 64, 130, 200), (129, 768, 5, 10),
    (128, 32, 3, 128), (50, 768, 2, 100), (1, 768, 1, 1), (4099, 384, 257, 20),
])
def test_dense_topk_matches_oracle(n, dim, b, k):
    corpus = synth.dense_corpus(n, dim, seed=n + dim)
    queries = synth.dense_queries(corpus, b, seed=b)
    _check(corpus, queries, k)


def test_dense_duplicate_rows_tie_order():
    corpus = synth.dense_corpus(4000, 768, seed=3, dup_frac=0.05)
    queries = synth.dense_queries(corpus, 40, seed=4, noise=0.2)
    _check(corpus, queries, 20, dups=True)
    # exact ties: identical rows must come out in descending row order
    import rag_hip
    idx = rag_hip.DenseIndex(np.repeat(corpus[:1], 300, axis=0))
    score, row, cnt = idx.search(corpus[:1], 50)
    assert list(row[0]) == list(range(299, 249, -1))
    assert np.all(score[0] == score[0, 0])


def test_dense_row_mask():
    corpus = synth.dense_corpus(6000, 768, seed=5)
    queries = synth.dense_queries(corpus, 70, seed=6)
    rng = np.random.default_rng(0)
    mask = (rng.random(6000) < 0.3).astype(np.uint8)
    _check(corpus, queries, 10, mask=mask)
    mask[:] = 0
    mask[[5, 77]] = 1
    _check(corpus, queries[:3], 10, mask=mask)


@pytest.mark.parametrize('metric', ['dot', 'euclidean'])
def test_dense_other_metrics(metric):
    rng = np.random.default_rng(11)
    corpus = (rng.standard_normal((3000, 128)) * rng.uniform(0.5, 2.0, (3000, 1))).astype(np.float32)
    queries = rng.standard_normal((40, 128)).astype(np.float32)
    import rag_hip
    idx = rag_hip.DenseIndex(corpus, metric=metric)
    score, row, cnt = idx.search(queries, 10)
    for b in range(40):
        ref_rows, ref_scores = dense_topk_ref(corpus, queries[b], 10, metric=metric)
        np.testing.assert_allclose(score[b], ref_scores, atol=2e-3 if metric == 'euclidean' else 1e-4 * 30, rtol=1e-5)
        assert np.mean(row[b] == ref_rows) >= 0.8


def test_dense_split_scores_stay_within_5e6_of_the_fp32_scan():
    """Split-bf16 products against the fp32 FMA chains on the same index: same rows except where two scores
    are closer than that, scores within 5e-6 (unit vectors, dim 768: the dropped lo.lo term is at most
    2^-16 |row||query| = 1.5e-5, reached only if every term had the same sign; near-duplicate queries,
    whose terms do, come to 1.3e-6); a dimension that is not a multiple of 32 refuses the split."""
    import rag_hip
    corpus = synth.dense_corpus(30000, 768, seed=21)
    queries = synth.dense_queries(corpus, 300, seed=22)
    idx = rag_hip.DenseIndex(corpus, split=True)
    s1, r1, c1 = idx.search(queries, 50)
    idx.set_split(False)
    s0, r0, c0 = idx.search(queries, 50)
    assert np.array_equal(c0, c1)
    assert float(np.abs(s1 - s0).max()) < 5e-6
    diff = r1 != r0
    assert diff.mean() < 0.01
    assert np.all(np.abs(s1 - s0)[diff] < 5e-6)
    idx.close()
    with pytest.raises(ValueError):
        rag_hip.DenseIndex(np.zeros((4, 48), np.float32), split=True)
    assert not rag_hip.DenseIndex(np.zeros((4, 48), np.float32)).split


def test_dense_wide_batches_take_the_same_rows_on_every_path(monkeypatch):
    """256 queries and more run the [256 x 256] direct-to-LDS scan, and over 262144 rows a threshold pre-pass in
    front of it (group maxima without a row mask, candidate lists with one).  Each of these can be switched off by
    an environment variable: the results must not change, and must match the oracle with a mask and for the
    Euclidean metric too."""
    import rag_hip
    n, dim, b, k = 270_000, 64, 300, 10
    corpus = synth.dense_corpus(n, dim, seed=31)
    queries = synth.dense_queries(corpus, b, seed=32)
    idx = rag_hip.DenseIndex(corpus)
    assert idx.split
    base = idx.search(queries, k)
    for var in ('RAG_DENSE_LIST_PREPASS', 'RAG_DENSE_NO_PREPASS', 'RAG_DENSE_NO_DMA', 'RAG_DENSE_RING4'):
        monkeypatch.setenv(var, '1')
        other = idx.search(queries, k)
        monkeypatch.delenv(var)
        for x, y in zip(base, other):
            assert np.array_equal(x, y), var
    rng = np.random.default_rng(1)
    mask = (rng.random(n) < 0.3).astype(np.uint8)
    _check_index(idx, corpus, queries[:260], k, 'cosine', mask, False)
    idx.close()
    _check(corpus[:20_000], queries[:257], k, metric='euclidean')
    _check(corpus[:100], queries[:256], 7)


@pytest.mark.parametrize('b,k', [(1, 100), (20, 10), (64, 100), (130, 37)])
def test_dense_two_level_strip_merge_equals_the_flat_merge(monkeypatch, b, k):
    """Small batches are scanned in many row strips; their lists are merged in groups of strips first and then
    across groups.  RAG_DENSE_MERGE_FLAT=1 merges every strip of a query in one go: same lists, bit for bit, in
    both forms of the scan, with a row mask and for distances."""
    import rag_hip
    n, dim = 150_000, 64
    corpus = synth.dense_corpus(n, dim, seed=41, dup_frac=0.01)
    queries = synth.dense_queries(corpus, b, seed=42)
    rng = np.random.default_rng(2)
    mask = (rng.random(n) < 0.5).astype(np.uint8)
    for metric in ('cosine', 'euclidean'):
        idx = rag_hip.DenseIndex(corpus, metric=metric)
        for split in (True, False):
            idx.set_split(split)
            for m in (None, mask):
                two = idx.search(queries, k, row_mask=m)
                monkeypatch.setenv('RAG_DENSE_MERGE_FLAT', '1')
                flat = idx.search(queries, k, row_mask=m)
                monkeypatch.delenv('RAG_DENSE_MERGE_FLAT')
                for x, y in zip(two, flat):
                    assert np.array_equal(x, y), (metric, split, m is not None)
        idx.close()


def test_dense_rescore_matches_scores():
    import rag_hip
    corpus = synth.dense_corpus(2000, 768, seed=8)
    queries = synth.dense_queries(corpus, 9, seed=9)
    idx = rag_hip.DenseIndex(corpus)
    rng = np.random.default_rng(1)
    cand = rng.integers(0, 2000, size=(9, 50))
    cand[0, 3] = -1
    got = idx.rescore(queries, cand)
    for b in range(9):
        ref = dense_scores_ref(corpus, queries[b])[np.maximum(cand[b], 0)]
        ref[cand[b] < 0] = 0
        np.testing.assert_allclose(got[b], ref, atol=TOL, rtol=0)


def test_dense_device_entry_point_and_row_base():
    import torch
    import rag_hip
    corpus = synth.dense_corpus(3000, 768, seed=12)
    queries = synth.dense_queries(corpus, 20, seed=13)
    dev = torch.device('cuda:0')
    rows_t = torch.from_numpy(corpus).to(dev)
    idx = rag_hip.DenseIndex(rows_t, row_base=100000, borrow=True)
    s, r, c = idx.search_dev(torch.from_numpy(queries).to(dev), 10)
    torch.cuda.synchronize()
    for b in range(20):
        ref_rows, ref_scores = dense_topk_ref(corpus, queries[b], 10)
        assert np.array_equal(r[b].cpu().numpy() - 100000, ref_rows)
        np.testing.assert_allclose(s[b].cpu().numpy(), ref_scores, atol=TOL, rtol=0)


def test_dense_argument_errors():
    import rag_hip
    corpus = synth.dense_corpus(100, 64, seed=1)
    idx = rag_hip.DenseIndex(corpus)
    with pytest.raises(rag_hip.RagHipError):
        idx.search(corpus[:1], rag_hip.MAX_K + 1)
    with pytest.raises(ValueError):
        idx.search(np.zeros((1, 32), np.float32), 5)
    with pytest.raises(rag_hip.RagHipError):
        rag_hip.DenseIndex(np.zeros((4, 6), np.float32))     # dim not a multiple of 4
