"""GPU: the 10M x 768 corpus of the eight-GPU configuration on ONE device, as eight logical row shards.

What the sharded path promises (each rank searches rows [g N/8, (g+1) N/8) with global row ids and
global BM25 statistics, the per-shard lists are merged) is checked at the configuration's full size:

  * dense: the merge of the eight shards' top-100 lists is the unsharded index's list, rows and scores
    bit for bit; both agree with a plain torch float32 product over all 10M rows (ids wherever
    neighbouring scores are 2e-6 apart, scores within 1e-5);
  * BM25: 10M documents, global idf / avgdl, document-partitioned postings; the merged lists equal the
    unsharded ones bit for bit, and for a few queries the scores equal a torch float64 evaluation of
    idf * impact summed in query-token order over the full posting lists, with nothing outside the
    list beating its last entry and ties by row descending.

Inputs are generated on the device (30.7 GB of rows): nothing here needs the host to hold the corpus.
"""
import numpy as np
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu
N, DIM, VOCAB, SHARDS, K = 10_000_000, 768, 200_000, 8, 100


def _rows_on_device(dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    rows = torch.empty((N, DIM), dtype=torch.float32, device=dev)
    step = 1 << 19
    for s in range(0, N, step):
        blk = torch.randn((min(step, N - s), DIM), generator=g, device=dev, dtype=torch.float32)
        rows[s:s + step] = torch.nn.functional.normalize(blk, dim=1)
    return rows, g


def test_dense_ten_million_rows_in_eight_logical_shards():
    import rag_hip
    dev = torch.device('cuda:0')
    free, _ = torch.cuda.mem_get_info()
    if free < 90 * (1 << 30):
        pytest.skip('needs 90 GB of free device memory')
    rows, g = _rows_on_device(dev, 11)
    b = 64
    pick = torch.randint(0, N, (b,), generator=g, device=dev)
    q = torch.randn((b, DIM), generator=g, device=dev)
    q[:48] = rows[pick[:48]] + 0.5 * q[:48] / DIM ** 0.5          # near neighbours of real rows + 16 random directions
    q = torch.nn.functional.normalize(q, dim=1).contiguous()

    full = rag_hip.DenseIndex(rows, borrow=True)
    want = [t.clone() for t in full.search_dev(q, K)]
    torch.cuda.synchronize()
    full.close()
    ss, rr, cc = [], [], []
    for s in range(SHARDS):
        lo, hi = N * s // SHARDS, N * (s + 1) // SHARDS
        shard = rag_hip.DenseIndex(rows[lo:hi], row_base=lo, borrow=True)
        x = shard.search_dev(q, K)
        torch.cuda.synchronize()
        assert int(x[1].min()) >= lo and int(x[1].max()) < hi        # global row ids of this shard's range
        ss.append(x[0].clone()); rr.append(x[1].clone()); cc.append(x[2].clone())
        shard.close()
    got = rag_hip.merge_topk_dev(torch.stack(ss, 1).contiguous(), torch.stack(rr, 1).contiguous(),
                                 torch.stack(cc, 1).contiguous(), K)
    torch.cuda.synchronize()
    assert bool((got[2] == K).all()) and bool((want[2] == K).all())
    assert torch.equal(got[1], want[1])
    assert torch.equal(got[0], want[0])                              # the same float32 chain per row
    assert bool((want[1][:48, 0] == pick[:48]).all())                # a perturbed row finds its source first

    # independent check over ALL rows: torch's float32 product
    sims = torch.empty((b, N), dtype=torch.float32, device=dev)
    step = 1 << 20
    for s in range(0, N, step):
        sims[:, s:s + step] = q @ rows[s:s + step].t()
    top = torch.topk(sims, K + 1, dim=1)
    assert float((top.values[:, :K] - want[0]).abs().max()) < 1e-5
    # a place of the list is decided when its score is 2e-6 away from both neighbours (float32 sums in another order
    # cannot be told apart below that): there the row ids must agree
    gap = top.values[:, :-1] - top.values[:, 1:] > 2e-6              # [b, K]: place j against place j + 1
    clear = gap.clone()
    clear[:, 1:] &= gap[:, :-1]
    assert float(clear.float().mean()) > 0.5                         # the id comparison is not vacuous
    assert torch.equal(top.indices[:, :K][clear], want[1][clear])
    assert bool((torch.sort(top.indices[:, :K], dim=1).values == torch.sort(want[1], dim=1).values).all(dim=1)[gap[:, -1]].all())
    rag_hip.check_faults()


def _sparse_on_device(dev, seed, mean_len=45.0, sigma=0.6, max_len=128):
    """The shape of synth.sparse_corpus (Zipf terms, log-normal document lengths), drawn on the device."""
    g = torch.Generator(device=dev).manual_seed(seed)
    ln = torch.empty((N,), dtype=torch.float32, device=dev).log_normal_(float(np.log(mean_len)), sigma, generator=g)
    ln = ln.round().clamp_(2, max_len).to(torch.int64)
    doc_off = torch.zeros((N + 1,), dtype=torch.int64, device=dev)
    doc_off[1:] = torch.cumsum(ln, 0)
    total = int(doc_off[-1])
    cdf = torch.from_numpy(np.cumsum(synth.zipf_probs(VOCAB))).to(dev)
    term = torch.empty((total,), dtype=torch.int32, device=dev)
    step = 1 << 26
    for s in range(0, total, step):
        u = torch.rand((min(step, total - s),), generator=g, device=dev, dtype=torch.float64)
        term[s:s + step] = torch.searchsorted(cdf, u).clamp_(max=VOCAB - 1).to(torch.int32)
    return term, doc_off, ln


def _index_from_stream(rag_hip, term, doc_off, ln, idf, avgdl, row_base):
    """term-id stream of a range of documents -> Bm25Index with the given (global) idf and avgdl; also the
    device-side CSR for the checker."""
    indptr, rws, tf = rag_hip.bm25_build_dev(term, doc_off, VOCAB)
    imp = rag_hip.bm25_impacts_dev(rws, tf, ln, avgdl, 1.5, 0.75)
    torch.cuda.synchronize()
    idx = rag_hip.Bm25Index(indptr.cpu().numpy(), rws.cpu().numpy().view(np.uint32), imp.cpu().numpy(), idf,
                            int(ln.shape[0]), row_base=row_base)
    return idx, (indptr, rws, imp)


def test_bm25_ten_million_documents_in_eight_logical_shards():
    import rag_hip
    from indexing.bm25_indexer import _idf
    dev = torch.device('cuda:0')
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * (1 << 30):
        pytest.skip('needs 60 GB of free device memory')
    term, doc_off, ln = _sparse_on_device(dev, 21)
    total = int(doc_off[-1])
    avgdl = total / N
    # global statistics from the unsharded build
    indptr, rws, tf = rag_hip.bm25_build_dev(term, doc_off, VOCAB)
    df = np.diff(indptr.cpu().numpy()).astype(np.int64)
    del rws, tf
    idf = _idf(df, N, VOCAB)
    full, (f_ptr, f_rows, f_imp) = _index_from_stream(rag_hip, term, doc_off, ln, idf, avgdl, 0)

    rng = np.random.default_rng(5)
    off_h = doc_off.cpu().numpy()
    queries = []
    cdf = np.cumsum(synth.zipf_probs(VOCAB))
    for i in range(64):
        d = int(rng.integers(0, N))
        own = term[off_h[d]:off_h[d + 1]].cpu().numpy()
        n = int(rng.integers(3, 13))
        qt = [int(own[rng.integers(0, own.shape[0])]) if rng.random() < 0.7
              else int(min(np.searchsorted(cdf, rng.random()), VOCAB - 1)) for _ in range(n)]
        queries.append(qt)
    queries[5] = [-1, -1, -1]                       # out of vocabulary: zero scores, highest rows first
    queries[6] = queries[7][:2] + queries[7][:2]    # repeated tokens count again
    qt_h, qo_h = rag_hip.Bm25Index.flatten(queries)
    qt, qo = torch.from_numpy(qt_h).to(dev), torch.from_numpy(qo_h).to(dev)
    want = [t.clone() for t in full.search_dev(qt, qo, int(qo_h[-1]), K)]
    torch.cuda.synchronize()
    full.close()

    ss, rr, cc = [], [], []
    for s in range(SHARDS):
        lo, hi = N * s // SHARDS, N * (s + 1) // SHARDS
        t_lo, t_hi = int(off_h[lo]), int(off_h[hi])
        shard, _ = _index_from_stream(rag_hip, term[t_lo:t_hi].contiguous(), (doc_off[lo:hi + 1] - t_lo).contiguous(),
                                      ln[lo:hi].contiguous(), idf, avgdl, lo)
        x = shard.search_dev(qt, qo, int(qo_h[-1]), K)
        torch.cuda.synchronize()
        ss.append(x[0].clone()); rr.append(x[1].clone()); cc.append(x[2].clone())
        shard.close()
    got = rag_hip.merge_topk_dev(torch.stack(ss, 1).contiguous(), torch.stack(rr, 1).contiguous(),
                                 torch.stack(cc, 1).contiguous(), K)
    torch.cuda.synchronize()
    assert bool((want[2] == K).all())
    assert torch.equal(got[2], want[2]) and torch.equal(got[1], want[1])
    assert torch.equal(got[0], want[0])                              # float64, bit for bit
    assert want[1][5].tolist() == list(range(N - 1, N - 1 - K, -1)) and bool((want[0][5] == 0).all())

    # independent evaluation of a few queries over the full posting lists: score[row] += idf * impact, one query
    # token after the other (each product and each sum rounded once, as numpy does it), then the ranking rule
    idf_d = torch.from_numpy(idf).to(dev)
    ptr_h = f_ptr.cpu().numpy()
    for i in (0, 6, 7, 33, 63):
        acc = torch.zeros((N,), dtype=torch.float64, device=dev)
        for t in queries[i]:
            if t < 0:
                continue
            a, z = int(ptr_h[t]), int(ptr_h[t + 1])
            r = f_rows[a:z].long() & 0xFFFFFFFF
            acc[r] = acc[r] + idf_d[t] * f_imp[a:z]                  # rows are distinct within a term
        rows_i, score_i = want[1][i], want[0][i]
        assert torch.equal(acc[rows_i], score_i)
        assert bool(((score_i[:-1] > score_i[1:]) | ((score_i[:-1] == score_i[1:]) & (rows_i[:-1] > rows_i[1:]))).all())
        last = score_i[-1]
        assert int((acc > last).sum()) == int((score_i > last).sum())            # nothing better was left out
        ties_in = rows_i[score_i == last]
        ties_all = torch.nonzero(acc == last)[:, 0]
        assert torch.equal(torch.sort(ties_all, descending=True).values[:ties_in.shape[0]], ties_in)
    rag_hip.check_faults()
