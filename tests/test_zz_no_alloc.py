"""GPU: the BM25 search that allocates nothing (rag_bm25_search_static_dev, through the C ABI) returns
bit for bit what rag_bm25_search_dev returns, on two calls in a row (the second smaller, so that it
finds the first call's leftovers in the static work array), and refuses sizes beyond the array."""
import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu


def _index(n_docs, vocab, seed):
    import rag_hip
    from indexing.bm25_indexer import build_arrays
    term, doc, ln = synth.sparse_corpus(n_docs, vocab, seed=seed, mean_len=30)
    term, _, v = synth.remap_first_appearance(term, vocab)
    arr = build_arrays(term, doc, ln, v, 1.5, 0.75)
    return rag_hip.Bm25Index(arr.indptr, arr.rows, arr.impacts(), arr.idf, n_docs), (term, doc, ln, v)


def _both(ix, queries, k):
    import torch
    import rag_hip
    dev = torch.device('cuda:0')
    terms, off = rag_hip.Bm25Index.flatten(queries)
    t, o = torch.from_numpy(terms).to(dev), torch.from_numpy(off).to(dev)
    b = len(queries)
    out = (torch.full((b, k), -7.0, dtype=torch.float64, device=dev), torch.full((b, k), -7, dtype=torch.int64, device=dev),
           torch.full((b,), -7, dtype=torch.int32, device=dev))
    got = ix.search_static_dev(t, o, int(off[-1]), k, out)
    want = ix.search_dev(t, o, int(off[-1]), k)
    torch.cuda.synchronize()
    return [x.cpu().numpy() for x in got], [x.cpu().numpy() for x in want]


def _same(got, want):
    gs, gr, gc = got
    ws, wr, wc = want
    assert np.array_equal(gc, wc)
    for q in range(gc.shape[0]):
        n = int(gc[q])
        assert np.array_equal(gr[q, :n], wr[q, :n]), q
        assert np.array_equal(gs[q, :n].view(np.uint64), ws[q, :n].view(np.uint64)), q     # fp64 bit for bit


def test_static_search_matches_search_dev_twice_in_a_row():
    import rag_hip
    rag_hip.init(0)
    n_docs, vocab = 150_000, 5000           # 74 document tiles: the sampled-threshold pass runs too
    ix, (term, doc, ln, v) = _index(n_docs, vocab, seed=11)
    big = synth.sparse_queries(term, doc, ln, v, 24, seed=12) + [[], [-1, -1], [3, 3, 3]]
    _same(*_both(ix, big, 100))
    small = synth.sparse_queries(term, doc, ln, v, 5, seed=13) + [[-1]]
    _same(*_both(ix, small, 10))
    _same(*_both(ix, big[:7], 1))


def test_static_search_small_corpus_and_zero_score_padding():
    ix, (term, doc, ln, v) = _index(300, 200, seed=21)
    qs = synth.sparse_queries(term, doc, ln, v, 9, seed=22) + [[-1, -1, -1]]
    _same(*_both(ix, qs, 50))
    _same(*_both(ix, qs[:2], 400))           # k beyond the corpus: every document comes back


def test_static_search_refuses_sizes_beyond_its_array():
    import torch
    import rag_hip
    ix, (term, doc, ln, v) = _index(150_000, 5000, seed=11)
    cap = rag_hip.lib().rag_bm25_search_static_capacity()
    assert cap == 32 << 20
    dev = torch.device('cuda:0')
    qs = synth.sparse_queries(term, doc, ln, v, 512, seed=3)     # 512 x 74 tiles x 100 x 12 B = 45 MB of lists
    terms, off = rag_hip.Bm25Index.flatten(qs)
    t, o = torch.from_numpy(terms).to(dev), torch.from_numpy(off).to(dev)
    out = (torch.full((512, 100), -7.0, dtype=torch.float64, device=dev),
           torch.full((512, 100), -7, dtype=torch.int64, device=dev), torch.full((512,), -7, dtype=torch.int32, device=dev))
    with pytest.raises(rag_hip.RagHipError, match='static array holds'):
        ix.search_static_dev(t, o, int(off[-1]), 100, out)
    torch.cuda.synchronize()
    assert int(out[2].min()) == -7 and int(out[1].min()) == -7          # outputs untouched
    want = ix.search_dev(t, o, int(off[-1]), 100)                       # the allocating form takes the same call
    torch.cuda.synchronize()
    assert int(want[2].min()) >= 0
