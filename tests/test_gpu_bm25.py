"""GPU parity: BM25 postings kernels (librag_hip via the C ABI) against the CPU oracle.
fp64 scores must be bit-identical; top-k lists identical (score desc, row desc)."""
import numpy as np
import pytest

from oracle.bm25_ref import BM25OkapiArrays, BM25OkapiLiteral, rank_rows_ref
from tests import synth

pytestmark = pytest.mark.gpu

WORDS = ("aspirin ibuprofen pain fever dose tablet daily mg side effect nausea headache liver kidney "
         "warning children adult pregnancy heart blood pressure the of and for with").split()


def _text_corpus(n, seed):
    rng = np.random.default_rng(seed)
    docs = []
    for _ in range(n):
        ln = int(rng.integers(0, 30))
        docs.append([WORDS[int(i)] for i in rng.integers(0, len(WORDS), ln)])
    for i in range(n):                # 'common' sits in 4 of 5 documents: negative raw idf
        if i % 5:
            docs[i].append('common')
    return docs


def _build_product(docs, k1=1.5, b=0.75):
    from indexing.bm25_indexer import BM25Indexer
    ix = BM25Indexer(k1=k1, b=b)
    chunks = [{'chunk_id': f'c{i}', 'doc_id': f'd{i}', 'source': 's', 'chunk_type': 't',
               'text': ' '.join(d), 'metadata': {}} for i, d in enumerate(docs)]
    ix.index_chunks(chunks)
    return ix


@pytest.mark.parametrize('n,seed', [(20, 0), (300, 1), (5000, 2)])
def test_bm25_scores_bit_identical_to_literal_form(n, seed):
    docs = _text_corpus(n, seed)
    lit = BM25OkapiLiteral(docs, k1=1.5, b=0.75)
    ix = _build_product(docs)
    a = ix.bm25
    assert a.avgdl == lit.avgdl
    for w, t in a.vocab.items():
        assert a.idf[t] == lit.idf[w], w
    queries = [['aspirin', 'pain'], ['the', 'the', 'of'], ['zzz_oov'], [], ['liver', 'zzz', 'liver', 'kidney', 'the'],
               ['dose'] * 7, ['common', 'heart']]
    for q in queries:
        got = ix._device_index.scores(ix.term_ids(q))
        want = lit.get_scores(q)
        assert got.dtype == np.float64 and np.array_equal(got, want), q


def test_bm25_negative_idf_floor_is_exercised():
    docs = _text_corpus(300, 5)
    lit = BM25OkapiLiteral(docs)
    assert lit.idf['common'] == 0.25 * lit.average_idf          # the epsilon floor replaced a negative idf
    ix = _build_product(docs)
    q = ['common', 'aspirin', 'common']
    assert np.array_equal(ix._device_index.scores(ix.term_ids(q)), lit.get_scores(q))


@pytest.mark.parametrize('n_docs,vocab,b,k', [(3000, 500, 40, 10), (20000, 5000, 64, 20), (9000, 2000, 30, 200),
                                              (4097, 300, 10, 1), (100, 50, 5, 100),
                                              (140000, 20000, 24, 50)])   # enough tiles for the sampled threshold
def test_bm25_topk_matches_oracle(n_docs, vocab, b, k):
    import rag_hip
    from indexing.bm25_indexer import build_arrays
    term, doc, ln = synth.sparse_corpus(n_docs, vocab, seed=n_docs)
    term, mapping, v = synth.remap_first_appearance(term, vocab)
    ref = BM25OkapiArrays.from_token_ids(term, doc, ln, v)
    arr = build_arrays(term, doc, ln, v, 1.5, 0.75)
    assert np.array_equal(arr.idf, ref.idf) and arr.avgdl == ref.avgdl
    idx = rag_hip.Bm25Index(arr.indptr, arr.rows, arr.impacts(), arr.idf, n_docs)
    queries = synth.sparse_queries(term, doc, ln, v, b, seed=b, oov_frac=0.1)
    queries.append([])                       # empty query: all scores zero
    queries.append([0, 0, 1])                # the most frequent terms, one repeated
    score, row, cnt = idx.search(queries, k)
    for i, q in enumerate(queries):
        want = ref.get_scores_ids(q)
        got_dense = idx.scores(q)
        assert np.array_equal(got_dense, want)
        order = rank_rows_ref(want, k)
        c = int(cnt[i])
        assert c == min(k, n_docs)
        assert np.array_equal(row[i, :c], order), (i, q)
        assert np.array_equal(score[i, :c], want[order])
        assert np.all(row[i, c:] == -1)
    idx.close()


def test_bm25_indexer_search_matches_oracle_dicts():
    from oracle.bm25_ref import bm25_search_ref
    from oracle.tokenize_ref import tokenize_ref
    docs = _text_corpus(400, 9)
    ix = _build_product(docs)
    for i, c in enumerate(ix.chunks):
        c['source'] = 'pubmed' if i % 3 == 0 else 'openfda'
    lit = BM25OkapiLiteral([tokenize_ref(c['text']) for c in ix.chunks])
    for query, top_k, filters in [("aspirin pain dose", 10, None), ("the liver", 5, {'source': 'pubmed'}),
                                  ("nothing matches here zzz", 6, None), ("heart", 4, {'source': 'nowhere'})]:
        want = bm25_search_ref(lit.get_scores(tokenize_ref(query)), ix.chunks, top_k=top_k, filters=filters)
        got = ix.search(query, top_k=top_k, filters=filters)
        assert got == want


def test_bm25_save_load_roundtrip(tmp_path):
    from indexing.bm25_indexer import BM25Indexer
    docs = _text_corpus(200, 4)
    ix = _build_product(docs)
    p = tmp_path / 'idx' / 'bm25_index.pkl'
    ix.save(str(p))
    ix2 = BM25Indexer()
    with pytest.raises(FileNotFoundError):
        ix2.load(str(tmp_path / 'missing.pkl'))
    ix2.load(str(p))
    assert ix2.search("aspirin pain", top_k=7) == ix.search("aspirin pain", top_k=7)
    assert ix2.get_stats() == ix.get_stats()
    with pytest.raises(ValueError):
        BM25Indexer().search("x")


def test_bm25_long_queries_and_dense_terms():
    """Queries of more than 32 tokens (the kernel looks posting ranges up 32 tokens at a time) and terms that
    nearly every document of a tile contains (more postings per tile than the kernel fetches ahead): scores and
    lists bit-identical to the oracle, repeated and out-of-vocabulary tokens included."""
    import rag_hip
    from indexing.bm25_indexer import build_arrays
    n_docs, vocab = 9000, 40                                  # 40 terms over 9000 documents: dense posting lists
    term, doc, ln = synth.sparse_corpus(n_docs, vocab, seed=77, mean_len=30)
    term, mapping, v = synth.remap_first_appearance(term, vocab)
    ref = BM25OkapiArrays.from_token_ids(term, doc, ln, v)
    arr = build_arrays(term, doc, ln, v, 1.5, 0.75)
    idx = rag_hip.Bm25Index(arr.indptr, arr.rows, arr.impacts(), arr.idf, n_docs)
    rng = np.random.default_rng(78)
    queries = []
    for n_tok in (31, 32, 33, 64, 65, 100, 3):
        q = rng.integers(0, v, size=n_tok).tolist()
        q[n_tok // 2] = -1                                     # one out-of-vocabulary token in the middle
        queries.append(q)
    queries.append([0] * 40)                                   # the most frequent term forty times
    score, row, cnt = idx.search(queries, 50)
    for i, q in enumerate(queries):
        want = ref.get_scores_ids(q)
        assert np.array_equal(idx.scores(q), want), i
        order = rank_rows_ref(want, 50)
        assert int(cnt[i]) == 50
        assert np.array_equal(row[i], order), i
        assert np.array_equal(score[i], want[order])
    idx.close()
