"""GPU: the device-side BM25 index build against the host (numpy) build of the same token streams."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _host(term, doc_off, vocab):
    """Counting on the host: the arrays indexing.bm25_indexer.build_arrays produces."""
    from indexing.bm25_indexer import build_arrays
    n_docs = doc_off.shape[0] - 1
    doc = np.repeat(np.arange(n_docs, dtype=np.int64), np.diff(doc_off))
    keep = term >= 0
    doc_len = np.diff(doc_off).astype(np.int64)
    return build_arrays(term[keep], doc[keep], doc_len, int(vocab), 1.5, 0.75)


def _device(term, doc_off, vocab):
    import rag_hip
    dev = torch.device('cuda:0')
    ip, rows, tf = rag_hip.bm25_build_dev(torch.from_numpy(term).to(dev), torch.from_numpy(doc_off).to(dev), vocab)
    return ip, rows, tf


def _case(n_docs, vocab, max_len, seed, zipf=True, with_ignored=False):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, max_len + 1, size=n_docs)
    doc_off = np.zeros(n_docs + 1, np.int64)
    doc_off[1:] = np.cumsum(lens)
    n = int(doc_off[-1])
    if zipf:
        p = 1.0 / np.arange(1, vocab + 1) ** 1.07
        term = rng.choice(vocab, size=n, p=p / p.sum()).astype(np.int32)
    else:
        term = rng.integers(0, vocab, size=n).astype(np.int32)
    if with_ignored and n:
        term[rng.random(n) < 0.05] = -1
    return term, doc_off


@pytest.mark.parametrize('n_docs,vocab,max_len,seed,zipf,ign', [
    (1, 5, 7, 0, False, False), (50, 30, 12, 1, True, False), (3000, 5000, 128, 2, True, False),
    (2000, 40, 200, 3, False, True), (7, 3, 3000, 4, False, False), (20000, 200000, 90, 5, True, True)])
def test_build_matches_host_counting(n_docs, vocab, max_len, seed, zipf, ign):
    term, doc_off = _case(n_docs, vocab, max_len, seed, zipf, ign)
    if doc_off[-1] == 0:
        term = np.zeros(0, np.int32)
    ref = _host(term, doc_off, vocab)
    ip, rows, tf = _device(term, doc_off, vocab)
    assert np.array_equal(ip.cpu().numpy(), ref.indptr)
    assert np.array_equal(rows.cpu().numpy().view(np.uint32), ref.rows)
    assert np.array_equal(tf.cpu().numpy().view(np.uint32), ref.tf)


def test_impacts_are_bit_identical_to_the_host_expression():
    import rag_hip
    term, doc_off = _case(4000, 3000, 100, 11)
    ref = _host(term, doc_off, 3000)
    dev = torch.device('cuda:0')
    ip, rows, tf = _device(term, doc_off, 3000)
    imp = rag_hip.bm25_impacts_dev(rows, tf, torch.from_numpy(ref.doc_len).to(dev), ref.avgdl, ref.k1, ref.b)
    assert np.array_equal(imp.cpu().numpy().view(np.uint64), ref.impacts().view(np.uint64))


def test_empty_documents_and_empty_corpus_of_tokens():
    doc_off = np.zeros(6, np.int64)
    ip, rows, tf = _device(np.zeros(0, np.int32), doc_off, 9)
    assert ip.cpu().tolist() == [0] * 10 and rows.numel() == 0 and tf.numel() == 0


def test_index_built_on_device_searches_like_the_host_built_one():
    """End to end: the CSR and impacts made on the device give the same top-k as the host-built index."""
    import rag_hip
    term, doc_off = _case(5000, 2000, 60, 21)
    ref = _host(term, doc_off, 2000)
    dev = torch.device('cuda:0')
    ip, rows, tf = _device(term, doc_off, 2000)
    imp = rag_hip.bm25_impacts_dev(rows, tf, torch.from_numpy(ref.doc_len).to(dev), ref.avgdl, ref.k1, ref.b)
    a = rag_hip.Bm25Index(ref.indptr, ref.rows, ref.impacts(), ref.idf, 5000)
    b = rag_hip.Bm25Index(ip.cpu().numpy(), rows.cpu().numpy().view(np.uint32), imp.cpu().numpy(), ref.idf, 5000)
    rng = np.random.default_rng(5)
    queries = [rng.integers(0, 2000, size=rng.integers(1, 9)).tolist() for _ in range(40)]
    sa, ra, ca = a.search(queries, 20)
    sb, rb, cb = b.search(queries, 20)
    assert np.array_equal(ca, cb) and np.array_equal(ra, rb) and np.array_equal(sa.view(np.uint64), sb.view(np.uint64))


def test_filter_mask_matches_host_evaluation():
    """Payload filters evaluated on the device from categorical codes against the host loop."""
    import rag_hip
    from indexing.vector_indexer import VectorIndexer

    class _Emb:
        embedding_dim, model_name = 8, 'none'
    rng = np.random.default_rng(3)
    n = 5003
    payloads = []
    for i in range(n):
        p = {'chunk_id': f'c{i}', 'doc_id': f'd{i // 7}', 'source': ['pubmed', 'openfda', 'kaggle'][int(rng.integers(0, 3))],
             'chunk_type': ['abstract', 'title'][i % 2], 'text': '', 'metadata': {'year': int(rng.integers(2000, 2004))}}
        if i % 11 == 0:
            del p['metadata']['year']
        payloads.append(p)
    vi = VectorIndexer(embedder=_Emb(), use_in_memory=True)
    vi.add_points(rng.standard_normal((n, 8)).astype(np.float32), payloads)
    for filt in ({'source': 'openfda'}, {'source': 'pubmed', 'chunk_type': 'title'}, {'metadata.year': 2002},
                 {'source': 'nowhere'}, {'doc_id': 'd100', 'chunk_type': 'abstract'}, {'metadata.year': 2001, 'source': 'kaggle'},
                 {'no_such_key': 1}, {'metadata': {'year': 2001}}):
        want = vi.row_mask(filt)
        got = vi.row_mask_dev(filt).cpu().numpy()
        assert np.array_equal(got, want), filt
    # and through the search: filtered top-k equals a search with the host-built mask
    q = rng.standard_normal((3, 8)).astype(np.float32)
    s1, r1, c1 = vi.search_rows(q, 10, {'source': 'pubmed', 'chunk_type': 'title'})
    s2, r2, c2 = vi._index.search(vi.prepare_queries(q), 10, row_mask=vi.row_mask({'source': 'pubmed', 'chunk_type': 'title'}))
    assert np.array_equal(r1, r2) and np.array_equal(c1, c2) and np.array_equal(s1, s2)
