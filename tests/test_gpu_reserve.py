"""GPU: rag_dense_reserve / rag_bm25_reserve size the handles' work buffers once; searches within the reserved bounds
then leave the device's free memory untouched (no hipMalloc / hipFree behind a `_dev` call) and return what an
unreserved handle returns."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_bytes():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def test_dense_searches_within_a_reservation_do_not_allocate():
    import rag_hip
    dev = torch.device('cuda:0')
    rag_hip.init(0)
    g = torch.Generator(device=dev).manual_seed(3)
    rows = torch.nn.functional.normalize(torch.randn((60_000, 128), generator=g, device=dev), dim=1)
    k = 50
    plain = rag_hip.DenseIndex(rows, metric='cosine', borrow=True)
    reserved = rag_hip.DenseIndex(rows, metric='cosine', borrow=True)
    reserved.reserve(300, k)
    batches = [1, 7, 32, 33, 64, 65, 200, 255, 256, 300]
    qs = {b: torch.nn.functional.normalize(torch.randn((b, 128), generator=g, device=dev), dim=1) for b in batches}
    outs = {b: (torch.empty((b, k), dtype=torch.float32, device=dev), torch.empty((b, k), dtype=torch.int64, device=dev),
                torch.empty((b,), dtype=torch.int32, device=dev)) for b in batches}
    want = {b: [t.clone() for t in plain.search_dev(qs[b], k)] for b in batches}
    free0 = _free_bytes()
    for b in batches + batches[::-1]:
        reserved.search_dev(qs[b], k, out=outs[b])
    assert _free_bytes() == free0
    for b in batches:
        assert torch.equal(outs[b][1], want[b][1]) and torch.equal(outs[b][0], want[b][0]) and torch.equal(outs[b][2], want[b][2])
    # a smaller k stays inside the reservation too
    small = (torch.empty((300, 10), dtype=torch.float32, device=dev), torch.empty((300, 10), dtype=torch.int64, device=dev),
             torch.empty((300,), dtype=torch.int32, device=dev))
    free0 = _free_bytes()
    reserved.search_dev(qs[300], 10, out=small)
    assert _free_bytes() == free0
    rag_hip.check_faults()
    plain.close()
    reserved.close()


def test_dense_reservation_covers_every_query_block_count_and_the_masked_pre_pass():
    """A plan's buffer sizes are not monotone in the batch size (strips are rounded per query-block count: three blocks
    of 256 queries need more workgroups than four), and a row mask switches the pre-pass to its list form: both stay
    inside a reservation made for the largest batch."""
    import rag_hip
    dev = torch.device('cuda:0')
    rag_hip.init(0)
    g = torch.Generator(device=dev).manual_seed(4)
    rows = torch.nn.functional.normalize(torch.randn((400_000, 64), generator=g, device=dev), dim=1)
    k = 20
    ix = rag_hip.DenseIndex(rows, metric='cosine', borrow=True)
    ix.reserve(1024, k)
    mask = (torch.rand((400_000,), generator=g, device=dev) < 0.7).to(torch.uint8)
    batches = [129, 257, 513, 600, 700, 768, 769, 900, 1024]
    qs = {b: torch.nn.functional.normalize(torch.randn((b, 64), generator=g, device=dev), dim=1) for b in batches}
    outs = {b: (torch.empty((b, k), dtype=torch.float32, device=dev), torch.empty((b, k), dtype=torch.int64, device=dev),
                torch.empty((b,), dtype=torch.int32, device=dev)) for b in batches}
    free0 = _free_bytes()
    for b in batches:
        ix.search_dev(qs[b], k, out=outs[b])
        ix.search_dev(qs[b], k, row_mask=mask, out=outs[b])
    assert _free_bytes() == free0
    # and the lists are right: the masked search of the last batch against torch
    s = qs[1024] @ rows.t()
    s[:, mask == 0] = -2.0
    want = torch.topk(s, k, dim=1).indices
    got = outs[1024][1]
    assert float((torch.sort(got, dim=1).values == torch.sort(want, dim=1).values).float().mean()) > 0.999
    rag_hip.check_faults()
    ix.close()


def test_bm25_searches_within_a_reservation_do_not_allocate():
    import rag_hip
    from tests import synth
    dev = torch.device('cuda:0')
    rag_hip.init(0)
    rng = np.random.default_rng(5)
    n_docs, vocab = 30_000, 5_000
    from indexing.bm25_indexer import build_arrays
    term, doc, ln = synth.sparse_corpus(n_docs, vocab, seed=11)
    term, _, v = synth.remap_first_appearance(term, vocab)
    arr = build_arrays(term, doc, ln, v, 1.5, 0.75)
    indptr, prow, impact, idf = arr.indptr, arr.rows, arr.impacts(), arr.idf
    vocab = v
    k = 40
    plain = rag_hip.Bm25Index(indptr, prow, impact, idf, n_docs)
    reserved = rag_hip.Bm25Index(indptr, prow, impact, idf, n_docs)
    reserved.reserve(128, 128 * 8, k)
    cases = []
    for b in (1, 16, 128):
        lens = rng.integers(1, 9, size=b)
        qoff = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).to(dev)
        qterms = torch.from_numpy(rng.integers(0, vocab, size=int(lens.sum())).astype(np.int32)).to(dev)
        out = (torch.empty((b, k), dtype=torch.float64, device=dev), torch.empty((b, k), dtype=torch.int64, device=dev),
               torch.empty((b,), dtype=torch.int32, device=dev))
        cases.append((qterms, qoff, int(lens.sum()), out, [t.clone() for t in plain.search_dev(qterms, qoff, int(lens.sum()), k)]))
    free0 = _free_bytes()
    for qterms, qoff, n, out, _ in cases + cases[::-1]:
        reserved.search_dev(qterms, qoff, n, k, out=out)
    assert _free_bytes() == free0
    for _, _, _, out, want in cases:
        assert all(torch.equal(a, b) for a, b in zip(out, want))
    rag_hip.check_faults()
    plain.close()
    reserved.close()
