"""CPU: host-side logic of the drop-in classes and scripts, with the GPU entry points swapped
for oracle-backed doubles (tests/fake_backend.py).  The real kernels are covered by the
``-m gpu`` tests; here the formatting, filters, persistence and plumbing are."""
import copy
import json
import os
import sys

import numpy as np
import pytest
import torch

from oracle import bm25_ref, fusion_ref, rerank_ref, tokenize_ref
from tests import fake_backend

HERE = os.path.dirname(__file__)
ROOT = os.path.dirname(HERE)
GOLD = json.load(open(os.path.join(HERE, 'golden', 'reference_pure_python.json')))

WORDS = ("aspirin ibuprofen pain fever dose tablet daily mg side effect nausea headache liver kidney "
         "warning children adult pregnancy heart blood pressure the of and for with").split()


def make_chunks(n, seed=0):
    rng = np.random.default_rng(seed)
    chunks = []
    for i in range(n):
        words = [WORDS[int(j)] for j in rng.integers(0, len(WORDS), int(rng.integers(3, 40)))]
        chunks.append({'chunk_id': f'chunk_{i % (n - n // 50)}', 'doc_id': f'doc_{i // 3}',
                       'source': ['pubmed', 'openfda', 'kaggle'][i % 3], 'chunk_type': 'abstract' if i % 2 else 'title',
                       'text': ' '.join(words), 'metadata': {'section': 's%d' % (i % 4), 'n': i}})
    return chunks


@pytest.fixture
def backend(monkeypatch):
    return fake_backend.install(monkeypatch)


@pytest.mark.parametrize('case', GOLD['tokenize'], ids=lambda c: repr(c['text'][:16]))
def test_tokenizer_matches_reference_fixture(case):
    from indexing.bm25_indexer import BM25Indexer
    assert BM25Indexer().tokenize(case['text']) == case['expected']


@pytest.mark.parametrize('case', GOLD['normalize'], ids=lambda c: '%s|%s|%s' % (c['text'][:16], bool(c['drug_mapping']),
                                                                            c['expand_abbreviations']))
def test_query_rewriting_matches_reference_fixture(case):
    """MedicalTermNormalizer.normalize_text / extract_medical_terms against the reference's own outputs
    (src/preprocessing/medical_term_normalizer.py; reached from scripts/query.py:122-136)."""
    from preprocessing import MedicalTermNormalizer
    nz = MedicalTermNormalizer()
    nz.drug_mapping = dict(case['drug_mapping'])
    assert nz.normalize_text(case['text'], expand_abbreviations=case['expand_abbreviations']) == case['expected']
    if case['terms'] is not None:
        assert nz.extract_medical_terms(case['text']) == case['terms']


def test_tokenizer_keeps_single_character_abbreviation_keys():
    from indexing.bm25_indexer import BM25Indexer
    from preprocessing import MedicalTermNormalizer
    norm = MedicalTermNormalizer()
    norm.abbreviations = dict(norm.abbreviations, K='potassium')
    ix = BM25Indexer(medical_normalizer=norm)
    text = "serum K 5 x level"
    assert ix.tokenize(text) == tokenize_ref.tokenize_ref(text, list(norm.abbreviations))
    assert 'k' in ix.tokenize(text) and 'x' not in ix.tokenize(text)


def test_native_tokenizer_and_vocabulary_match_the_python_tokenizer():
    """rag_bm25_tokenize_* (host C++ in librag_hip.so, no GPU): ASCII documents tokenised natively, the others handed
    over as tokens; term ids, document lengths and first-appearance vocabulary equal what the Python loop gives on
    the reference's tokenizer fixture and on random ASCII text with punctuation, digits, underscores and control
    characters."""
    import rag_hip
    from indexing.bm25_indexer import BM25Indexer
    from preprocessing import MedicalTermNormalizer
    norm = MedicalTermNormalizer()
    norm.abbreviations = dict(norm.abbreviations, K='potassium', x='times')
    ix = BM25Indexer(medical_normalizer=norm)
    rng = np.random.default_rng(7)
    alphabet = list("abcXYZ019_ -.,;:/()\t\n\x00'\"?!kKxX")
    texts = [c['text'] for c in GOLD['tokenize']] + ['serum K 5 x level', 'K', 'x', '_', 'a_b c__d 9 99', '']
    texts += [''.join(alphabet[int(j)] for j in rng.integers(0, len(alphabet), int(rng.integers(0, 80)))) for _ in range(300)]
    want_tokens = [ix.tokenize(t) for t in texts]          # the Python tokenizer (itself pinned by the reference's fixture)
    vocab, want_ids = {}, []
    for doc in want_tokens:
        want_ids.append([vocab.setdefault(w, len(vocab)) for w in doc])
    docs, flags = [], []
    for t in texts:
        if t.isascii():
            docs.append(t.encode('ascii'))
            flags.append(0)
        else:
            docs.append('\n'.join(ix.tokenize(t)).encode('utf-8'))
            flags.append(1)
    assert 0 < sum(flags) < len(flags)
    ids, doc_len, words = rag_hip.bm25_tokenize(docs, flags, ix._short_key_string())
    assert words == list(vocab)
    assert doc_len.tolist() == [len(d) for d in want_tokens]
    assert ids.tolist() == [t for d in want_ids for t in d]
    # a non-ASCII document that is not marked as tokenised is refused, not mangled
    with pytest.raises(rag_hip.RagHipError, match='non-ASCII'):
        rag_hip.bm25_tokenize(['caf\u00e9'.encode('utf-8')], [0])
    # empty corpus
    ids, doc_len, words = rag_hip.bm25_tokenize([], [])
    assert ids.shape == (0,) and doc_len.shape == (0,) and words == []


def test_bm25_indexer_matches_oracle(backend, tmp_path):
    from indexing.bm25_indexer import BM25Indexer
    chunks = make_chunks(300)
    ix = BM25Indexer(k1=1.2, b=0.6)
    assert ix.get_stats() == {'error': 'Index not built'}
    with pytest.raises(ValueError, match='Index not built'):
        ix.search('aspirin')
    ix.index_chunks(chunks)
    lit = bm25_ref.BM25OkapiLiteral([tokenize_ref.tokenize_ref(c['text']) for c in chunks], k1=1.2, b=0.6)
    for query, k, filt in [('aspirin pain', 10, None), ('liver the the', 7, {'source': 'openfda'}),
                           ('zzz unknown', 5, None), ('heart', 3, {'source': 'none'}), ('dose', 0, None)]:
        want = bm25_ref.bm25_search_ref(lit.get_scores(tokenize_ref.tokenize_ref(query)), chunks, top_k=k,
                                        filters=filt) if k else []
        assert ix.search(query, top_k=k, filters=filt) == want
    stats = ix.get_stats()
    assert stats['num_documents'] == 300 and stats['k1'] == 1.2 and stats['b'] == 0.6
    assert stats['vocab_size'] == len(lit.idf) and stats['avg_doc_length'] == float(np.mean(lit.doc_len))
    # rebuilding does not append to the previous corpus
    ix.index_chunks(chunks[:50])
    assert len(ix.tokenized_corpus) == 50
    # persistence: the reference's five keys, reloadable
    p = tmp_path / 'sub' / 'bm25_index.pkl'
    ix.save(str(p))
    import pickle
    with open(p, 'rb') as f:
        assert sorted(pickle.load(f)) == ['b', 'bm25', 'chunks', 'k1', 'tokenized_corpus']
    ix2 = BM25Indexer()
    ix2.load(str(p))
    assert (ix2.k1, ix2.b) == (1.2, 0.6)
    assert ix2.search('aspirin pain', top_k=5) == ix.search('aspirin pain', top_k=5)
    with pytest.raises(FileNotFoundError):
        ix2.load(str(tmp_path / 'nope.pkl'))


def test_bm25_load_reads_a_reference_written_pickle_without_running_its_code(backend, tmp_path, monkeypatch):
    """The reference pickles {'bm25': <rank_bm25.BM25Okapi instance>, 'chunks', 'tokenized_corpus', 'k1', 'b'}
    (src/indexing/bm25_indexer.py:170-176).  rank_bm25 is not installed: the loader admits that class (and the
    chunker's Chunk objects) only as inert attribute bags, rebuilds the index from the token lists, and refuses any
    other global - nothing in the file is executed."""
    import pickle
    import sys
    import types
    from indexing.bm25_indexer import BM25Indexer
    chunks = make_chunks(120)
    ours = BM25Indexer(k1=1.3, b=0.7)
    ours.index_chunks(chunks)

    # a stand-in for the upstream class, present only while the file is written
    mod = types.ModuleType('rank_bm25')

    class BM25Okapi:
        def __init__(self, corpus, k1, b):
            self.k1, self.b, self.epsilon, self.corpus_size = k1, b, 0.25, len(corpus)
            self.doc_len = [len(d) for d in corpus]
            self.avgdl = sum(self.doc_len) / max(len(corpus), 1)
            self.doc_freqs = [{w: d.count(w) for w in set(d)} for d in corpus]
            self.idf = {w: 0.5 for d in corpus for w in d}
            self.tokenizer = None

        def __reduce_ex__(self, protocol):       # what object.__reduce_ex__ gives a plain instance
            return object.__reduce_ex__(self, protocol)
    BM25Okapi.__module__, BM25Okapi.__qualname__ = 'rank_bm25', 'BM25Okapi'
    mod.BM25Okapi = BM25Okapi
    chunker = types.ModuleType('src.preprocessing.chunker')

    class Chunk:
        def __init__(self, **kw):
            self.__dict__.update(kw)
    Chunk.__module__, Chunk.__qualname__ = 'src.preprocessing.chunker', 'Chunk'
    chunker.Chunk = Chunk
    monkeypatch.setitem(sys.modules, 'rank_bm25', mod)
    for parent in ('src', 'src.preprocessing'):
        if parent not in sys.modules:
            monkeypatch.setitem(sys.modules, parent, types.ModuleType(parent))
    monkeypatch.setitem(sys.modules, 'src.preprocessing.chunker', chunker)
    toks = [ours.tokenize(c['text']) for c in chunks]
    objs = [Chunk(**c) if i % 2 else c for i, c in enumerate(chunks)]        # both chunk spellings
    p = tmp_path / 'bm25_index.pkl'
    with open(p, 'wb') as f:
        pickle.dump({'bm25': BM25Okapi(toks, 1.3, 0.7), 'chunks': objs, 'tokenized_corpus': toks, 'k1': 1.3, 'b': 0.7}, f)
    monkeypatch.delitem(sys.modules, 'rank_bm25')
    monkeypatch.delitem(sys.modules, 'src.preprocessing.chunker')
    with pytest.raises(ModuleNotFoundError):
        with open(p, 'rb') as f:
            pickle.load(f)                       # the plain loader needs the package
    ix = BM25Indexer()
    ix.load(str(p))
    assert (ix.k1, ix.b) == (1.3, 0.7) and len(ix.chunks) == 120 and all(isinstance(c, dict) for c in ix.chunks)
    for q in ('aspirin pain', 'liver the the', 'zzz'):
        assert ix.search(q, top_k=7) == ours.search(q, top_k=7)

    # a file that names any other global is refused before anything runs
    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ('echo should-never-run > %s' % (tmp_path / 'pwned'),))
    bad = tmp_path / 'bad.pkl'
    with open(bad, 'wb') as f:
        pickle.dump({'bm25': Evil(), 'chunks': [], 'tokenized_corpus': [], 'k1': 1.5, 'b': 0.75}, f)
    with pytest.raises(pickle.UnpicklingError, match='does not admit'):
        BM25Indexer().load(str(bad))
    assert not (tmp_path / 'pwned').exists()
    other = tmp_path / 'other.pkl'
    with open(other, 'wb') as f:
        pickle.dump([1, 2, 3], f)
    with pytest.raises(ValueError, match='not a BM25 index file'):
        BM25Indexer().load(str(other))


def test_bm25_accepts_chunk_objects(backend):
    from indexing.bm25_indexer import BM25Indexer
    from preprocessing import Chunk
    objs = [Chunk(chunk_id=f'c{i}', doc_id='d', source='s', chunk_type='t', text=t, metadata={'i': i})
            for i, t in enumerate(['aspirin pain relief', 'liver warning', 'aspirin aspirin dose'])]
    ix = BM25Indexer()
    ix.index_chunks(objs)
    got = ix.search('aspirin', top_k=2)
    assert [g['chunk_id'] for g in got] == ['c2', 'c0'] and got[0]['metadata'] == {'i': 2}


def _embedder():
    from indexing import MedicalEmbedder
    return MedicalEmbedder(model_name='random:tiny', batch_size=7, device='cpu')


def test_embedder_contract():
    e = _embedder()
    assert e.embedding_dim == 64 and e.get_model_info()['model_name'] == 'random:tiny'
    texts = ['aspirin for pain', 'a', 'liver ' * 300, '']
    out = e.encode(texts)
    assert out.shape == (4, 64) and out.dtype == np.float32 and out.flags['C_CONTIGUOUS']
    assert np.allclose(np.linalg.norm(out, axis=1), 1, atol=1e-5)
    # order independent of batching / length sorting, single string accepted
    assert np.allclose(e.encode('a')[0], out[1], atol=1e-6)
    assert np.allclose(e.encode(list(reversed(texts)))[::-1], out, atol=1e-6)
    raw = e.encode(texts[:2], normalize=False)
    assert not np.allclose(np.linalg.norm(raw, axis=1), 1, atol=1e-3)
    assert len(e.tokenize('liver ' * 2000)) == e.max_length
    assert np.allclose(e.encode_chunks([{'text': 'a'}])[0], out[1], atol=1e-6)
    with pytest.raises(FileNotFoundError):
        from indexing import MedicalEmbedder
        MedicalEmbedder(model_name='some/model-that-is-not-cached', device='cpu')


def test_vector_indexer_matches_oracle(backend, tmp_path):
    from indexing import VectorIndexer
    from oracle.dense_ref import dense_topk_ref, normalize_rows_ref
    chunks = make_chunks(120, seed=3)
    e = _embedder()
    vi = VectorIndexer(collection_name='col', embedder=e, storage_path=str(tmp_path / 'vdb'))
    assert 'error' in vi.get_stats()
    vi.create_collection(force_recreate=True)
    vi.index_chunks(chunks)
    emb = normalize_rows_ref(e.encode([c['text'] for c in chunks]))
    q = 'aspirin side effect'
    qv = e.encode(q)[0]
    rows, scores = dense_topk_ref(emb, qv, 8)
    got = vi.search(q, top_k=8)
    assert [g['chunk_id'] for g in got] == [chunks[r]['chunk_id'] for r in rows]
    assert np.allclose([g['score'] for g in got], scores, atol=1e-6)
    assert set(got[0]) == {'chunk_id', 'doc_id', 'source', 'chunk_type', 'text', 'metadata', 'score'}
    # payload filters: every key must match; dotted keys reach into metadata; unknown key matches nothing
    mask = np.array([c['source'] == 'pubmed' and c['metadata']['section'] == 's1' for c in chunks], np.uint8)
    rows_f, _ = dense_topk_ref(emb, qv, 5, row_mask=mask)
    got = vi.search(q, top_k=5, filters={'source': 'pubmed', 'metadata.section': 's1'})
    assert [g['metadata']['n'] for g in got] == list(rows_f)
    assert vi.search(q, top_k=5, filters={'no_such_key': 1}) == []
    # score threshold cuts the tail
    thr = float(scores[3])
    assert len(vi.search(q, top_k=8, score_threshold=thr)) == int(np.sum(scores >= thr))
    stats = vi.get_stats()
    assert stats == {'collection_name': 'col', 'points_count': 120, 'embedding_dim': 64,
                     'model_name': 'random:tiny', 'storage_path': str(tmp_path / 'vdb')}
    # a second indexer attaches to the persisted collection
    vi2 = VectorIndexer(collection_name='col', embedder=e, storage_path=str(tmp_path / 'vdb'))
    assert vi2.search(q, top_k=8) == vi.search(q, top_k=8)
    assert np.array_equal(vi2._canon, vi._canon) and vi._canon[119] != 119       # duplicate chunk ids
    vi2.delete_collection()
    assert 'error' in VectorIndexer(collection_name='col', embedder=e, storage_path=str(tmp_path / 'vdb')).get_stats()
    mem = VectorIndexer(collection_name='m', embedder=e, use_in_memory=True)
    mem.create_collection()
    mem.index_chunks(chunks[:10])
    assert mem.get_stats()['storage_path'] == 'in-memory' and len(mem.search(q, top_k=20)) == 10


@pytest.mark.parametrize('name', sorted(GOLD['rrf']))
def test_hybrid_rrf_matches_reference_fixture(backend, name, tmp_path):
    from indexing import HybridIndexer
    c = GOLD['rrf'][name]
    h = HybridIndexer(storage_path=str(tmp_path))
    assert h._reciprocal_rank_fusion(copy.deepcopy(c['vector']), copy.deepcopy(c['bm25']), top_k=c['top_k']) == c['expected']


@pytest.mark.parametrize('name', sorted(GOLD['weighted']))
def test_hybrid_weighted_matches_reference_fixture(backend, name, tmp_path):
    from indexing import HybridIndexer
    c = GOLD['weighted'][name]
    h = HybridIndexer(storage_path=str(tmp_path))
    v, b = copy.deepcopy(c['vector']), copy.deepcopy(c['bm25'])
    got = h._weighted_fusion(v, b, vector_weight=c['vector_weight'], bm25_weight=c['bm25_weight'], top_k=c['top_k'])
    assert got == c['expected'] and v == c['vector_after'] and b == c['bm25_after']


def test_hybrid_search_matches_oracle(backend, tmp_path):
    from indexing import HybridIndexer, HybridRetriever, VectorIndexer, BM25Indexer
    chunks = make_chunks(200, seed=5)
    e = _embedder()
    h = HybridRetriever(VectorIndexer('c', e, str(tmp_path / 'v')), BM25Indexer(), storage_path=str(tmp_path / 'i'))
    h.build_index(chunks)
    for method in ('rrf', 'weighted'):
        for filt in (None, {'source': 'kaggle'}):
            want = fusion_ref.hybrid_search_ref(h.vector_indexer.search, h.bm25_indexer.search, 'liver warning dose',
                                                top_k=6, vector_weight=0.7, bm25_weight=0.3, filters=filt,
                                                fusion_method=method)
            got = h.query('liver warning dose', top_k=6, vector_weight=0.7, bm25_weight=0.3, filters=filt,
                          fusion_method=method)
            assert got == want and got[0]['fusion_method'] == method
    # a missing retriever contributes an empty list
    only_b = HybridIndexer(None, h.bm25_indexer, storage_path=str(tmp_path / 'i'))
    got = only_b.search('aspirin', top_k=4)
    assert [g['chunk_id'] for g in got] == [g['chunk_id'] for g in h.bm25_indexer.search('aspirin', top_k=8)][:4]
    assert set(h.get_stats()) == {'vector', 'bm25'} and set(only_b.get_stats()) == {'bm25'}
    h.save_bm25()
    assert (tmp_path / 'i' / 'bm25_index.pkl').exists()
    h.load_bm25()
    assert HybridIndexer(storage_path=str(tmp_path / 'x')).search('q') == []


@pytest.mark.parametrize('name', sorted(GOLD['rerank']))
def test_base_reranker_matches_reference_fixture(backend, name):
    from reranker import BaseReranker
    c = GOLD['rerank'][name]

    class Table(BaseReranker):
        def score_pairs(self, pairs):
            self.pairs = list(pairs)
            return c['scores'][:len(self.pairs)]
    rr = Table(top_n=c['top_n'])
    got = rr.rerank('the query', copy.deepcopy(c['results']), top_k=c['top_k'])
    assert got == c['expected']
    if c['pairs'] is not None:
        assert [list(p) for p in rr.pairs] == [list(p) for p in c['pairs']]


def test_rerankers_and_factory(backend):
    from reranker import build_reranker, SimpleSimilarityReranker, CrossEncoderReranker, BaseReranker
    e = _embedder()
    assert build_reranker('none') is None
    simple = build_reranker('simple', top_n=4, embedder=e)
    assert isinstance(simple, SimpleSimilarityReranker) and simple.top_n == 4 and simple.dim == 64
    results = [{'chunk_id': f'c{i}', 'text': t, 'score': 0.1, 'fusion_method': 'rrf'}
               for i, t in enumerate(['aspirin pain', 'liver damage warning', 'aspirin dose daily', 'kidney', 'x y z'])]
    out = simple.rerank('aspirin dose', copy.deepcopy(results), top_k=3)
    assert len(out) == 3 and all(r['fusion_method'] == 'rrf+rerank' for r in out)
    assert [r['rerank_score'] for r in out] == sorted((r['rerank_score'] for r in out), reverse=True)
    assert {r['chunk_id'] for r in out} <= {'c0', 'c1', 'c2', 'c3'}          # pool is the first top_n
    vec = e.encode(['aspirin dose'] + [r['text'] for r in results[:4]])
    want = [float(rerank_ref.cosine_ref(vec[0].tolist(), v.tolist())) for v in vec[1:]]
    got = simple.score_pairs([('aspirin dose', r['text']) for r in results[:4]])
    assert np.allclose(got, want, atol=1e-5)
    ce = CrossEncoderReranker(model_name='random:tiny', top_n=5, device='cpu')
    scores = ce.score_pairs([('aspirin dose', r['text']) for r in results])
    assert len(scores) == 5 and all(isinstance(s, float) for s in scores)
    assert np.allclose(ce.score_pairs([('aspirin dose', results[2]['text'])])[0], scores[2], atol=1e-5)
    ids, types = ce.encode_pair('a b', 'c d e')
    assert ids[0] == 101 and ids.count(102) == 2 and types == [0, 0, 0, 0, 1, 1, 1, 1]
    with pytest.raises(FileNotFoundError):          # no silent switch to another reranker
        build_reranker('crossencoder', cross_model='cross-encoder/not-in-cache')
    assert BaseReranker().rerank('q', []) == []


def test_simple_reranker_from_resident_rows_equals_re_embedding(backend):
    """SimpleSimilarityReranker(vector_indexer=...) scores pool texts from the stored embeddings; texts the
    collection does not hold are embedded as before.  Same scores, same order (reference rerank.py:46-68)."""
    from indexing import VectorIndexer
    from reranker import build_reranker, SimpleSimilarityReranker
    e = _embedder()
    chunks = make_chunks(200, seed=5)
    vi = VectorIndexer('c', e, use_in_memory=True)
    vi.create_collection()
    vi.index_chunks(chunks)
    plain = SimpleSimilarityReranker(embedder=e, top_n=30)
    fast = build_reranker('simple', top_n=30, embedder=e, vector_indexer=vi)
    assert fast.vector_indexer is vi
    q = 'aspirin dose for children'
    results = vi.search(q, top_k=25) + [{'chunk_id': 'foreign', 'text': 'a text the collection never saw', 'score': 0.0}]
    pairs = [(q, r['text']) for r in results]
    assert vi.rows_of_texts([r['text'] for r in results])[-1] == -1
    a, b = plain.score_pairs(pairs), fast.score_pairs(pairs)
    assert np.allclose(a, b, atol=2e-6)
    ra, rb = plain.rerank(q, copy.deepcopy(results), top_k=10), fast.rerank(q, copy.deepcopy(results), top_k=10)
    assert [r['chunk_id'] for r in ra] == [r['chunk_id'] for r in rb]
    # vectors handed in from outside are not known to be this embedder's embeddings of the texts: the reranker then
    # embeds the texts itself, as the reference does (rerank.py:60-68), instead of scoring from the stored rows
    assert vi.rows_are_embeddings
    foreign = VectorIndexer('f', e, use_in_memory=True)
    foreign.create_collection()
    rng = np.random.default_rng(0)
    foreign.add_points(rng.standard_normal((len(chunks), e.embedding_dim)).astype(np.float32),
                       [dict(c, chunk_id=c['chunk_id']) for c in chunks])
    assert not foreign.rows_are_embeddings
    c = SimpleSimilarityReranker(embedder=e, top_n=30, vector_indexer=foreign).score_pairs(pairs)
    assert np.allclose(a, c, atol=2e-6)               # the reference's cosines, not dot products with the random rows
    # a collection of another metric or embedder is not used
    other = VectorIndexer('d', e, use_in_memory=True, distance_metric='dot')
    assert build_reranker('simple', embedder=e, vector_indexer=other).vector_indexer is None
    with pytest.raises(ValueError):
        SimpleSimilarityReranker(embedder=e, vector_indexer=other)


def test_scripts_build_and_query_1k_chunks_bm25_only(backend, tmp_path, capsys):
    """The smallest benchmark configuration: 1k synthetic chunks, BM25-only top_k=5 through the scripts."""
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import build_index
    import query as query_script
    chunks = make_chunks(1000, seed=11)
    cp = tmp_path / 'chunks.jsonl'
    with open(cp, 'w') as f:
        for c in chunks:
            f.write(json.dumps(c) + '\n')
    out_dir, vdb = tmp_path / 'indices', tmp_path / 'vdb'
    args = build_index.build_parser().parse_args([
        '--chunks_path', str(cp), '--embedding_model', 'random:tiny', '--output_dir', str(out_dir),
        '--vector_storage_path', str(vdb), '--drug_mapping_path', str(tmp_path / 'none.json')])
    build_index.main(args)
    meta = json.load(open(out_dir / 'index_metadata.json'))
    assert meta['num_chunks'] == 1000 and meta['embedding_dim'] == 64 and meta['collection_name'] == 'medical_rag'
    assert meta['bm25_params'] == {'k1': 1.5, 'b': 0.75} and meta['distance_metric'] == 'cosine'
    assert (out_dir / 'bm25_index.pkl').exists() and (vdb / 'medical_rag' / 'embeddings.f32').exists()

    q = 'What are the side effects of aspirin for pt w/ HTN?'
    a = query_script.build_parser().parse_args(['--query', q, '--indices_dir', str(out_dir), '--top_k', '5',
                                                '--output', str(tmp_path / 'out.json')])
    a.filters = None
    hybrid = query_script.load_hybrid_indexer(str(out_dir), meta)
    # BM25-only: ids identical to the oracle on the same chunks
    lit = bm25_ref.BM25OkapiLiteral([tokenize_ref.tokenize_ref(c['text']) for c in chunks])
    want = bm25_ref.bm25_search_ref(lit.get_scores(tokenize_ref.tokenize_ref(q)), chunks, top_k=5)
    assert hybrid.bm25_indexer.search(q, top_k=5) == want
    # full CLI path (hybrid + simple rerank), output schema of the reference
    a.reranker_kind, a.rerank_top_n = 'simple', 20
    out = query_script.main(a)
    saved = json.load(open(tmp_path / 'out.json'))
    assert sorted(saved) == ['num_results', 'preprocessed', 'query', 'results'] and saved['query'] == q
    assert saved['num_results'] == 5 and len(saved['results']) == 5
    assert all(r['fusion_method'] == 'rrf+rerank' and 'rerank_score' in r for r in saved['results'])
    assert sorted(saved['preprocessed']) == ['cleaned', 'entities', 'medical_terms', 'normalized', 'original']
    assert 'hypertension' in saved['preprocessed']['normalized']
    # missing index directory: message, no exception
    a.indices_dir = str(tmp_path / 'nowhere')
    assert query_script.main(a) is None


# ---- tile lists of the fused projection + attention kernel (host functions of the library) --------------------
def test_qkv_attention_tiles_and_pack_on_the_host():
    import numpy as np
    import rag_hip
    rng = np.random.default_rng(9)
    lens = np.concatenate([rng.integers(1, 257, size=300), [256, 256, 1, 1, 0, 0, 128, 129, 127]]).astype(np.int32)
    # as they come: greedy, in order (empty sequences are not allowed there)
    live = lens[lens > 0]
    cu = np.concatenate([[0], np.cumsum(live)]).astype(np.int64)
    tiles = rag_hip.enc_qkv_attention_tiles(cu)
    assert tiles[0] == 0 and tiles[-1] == len(live) and np.all(np.diff(tiles) > 0)
    tok = cu[tiles]
    assert int(np.diff(tok).max()) <= 256
    for t in range(len(tiles) - 2):                          # greedy: the next sequence would not have fitted
        assert tok[t + 1] - tok[t] + live[tiles[t + 1]] > 256
    # packed: a permutation, empty sequences last and in no tile, tiles nearly full
    order, ptiles = rag_hip.enc_qkv_attention_pack(lens)
    assert sorted(order.tolist()) == list(range(len(lens)))
    n_live = int((lens > 0).sum())
    assert ptiles[-1] == n_live and np.all(lens[order[n_live:]] == 0) and np.all(lens[order[:n_live]] > 0)
    pcu = np.concatenate([[0], np.cumsum(lens[order])])
    ptok = pcu[ptiles]
    assert int(np.diff(ptok).max()) <= 256 and np.all(np.diff(ptiles) > 0)
    assert len(ptiles) - 1 <= len(tiles) - 1                 # never more tiles than taking them as they come
    assert len(ptiles) - 1 <= int(np.ceil(pcu[-1] / 256 * 1.12))      # lengths uniform in 1..256: many cannot be paired
    for t in range(len(ptiles) - 1):                         # every tile starts with its longest sequence
        seg = lens[order[ptiles[t]:ptiles[t + 1]]]
        assert seg[0] == seg.max()
    # deterministic, and stable inside a length
    order2, _ = rag_hip.enc_qkv_attention_pack(lens)
    assert np.array_equal(order, order2)
    # limits
    assert rag_hip.enc_qkv_attention_pack(np.array([5, 257], np.int32)) is None
    with pytest.raises(rag_hip.RagHipError):
        rag_hip.enc_qkv_attention_tiles(np.array([0, 10, 10], np.int64))      # an empty sequence
    o, t = rag_hip.enc_qkv_attention_pack(np.zeros(0, np.int32))
    assert len(o) == 0 and t.tolist() == [0]
