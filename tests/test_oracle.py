"""CPU: the oracle against the golden fixtures (outputs of the reference's own pure-Python
functions and scores stored in its result files) and against itself."""
import copy
import json
import os

import numpy as np
import pytest

from oracle import bm25_ref, dense_ref, fusion_ref, rerank_ref, tokenize_ref
from tests import synth

HERE = os.path.dirname(__file__)
GOLD = json.load(open(os.path.join(HERE, 'golden', 'reference_pure_python.json')))
STORED = json.load(open(os.path.join(HERE, 'golden', 'results_rrf_scores.json')))['scores']


def test_abbreviation_table_matches_reference():
    assert list(tokenize_ref.ABBREVIATION_KEYS) == GOLD['abbreviation_keys']


@pytest.mark.parametrize('case', GOLD['tokenize'], ids=lambda c: repr(c['text'][:20]))
def test_tokenizer_matches_reference(case):
    assert tokenize_ref.tokenize_ref(case['text']) == case['expected']


@pytest.mark.parametrize('name', sorted(GOLD['rrf']))
def test_rrf_matches_reference(name):
    c = GOLD['rrf'][name]
    got = fusion_ref.rrf_ref(copy.deepcopy(c['vector']), copy.deepcopy(c['bm25']), top_k=c['top_k'], k=c['k'])
    assert got == c['expected']
    assert [float(g['score']).hex() for g in got] == [float(e['score']).hex() for e in c['expected']]


@pytest.mark.parametrize('name', sorted(GOLD['weighted']))
def test_weighted_matches_reference(name):
    c = GOLD['weighted'][name]
    v, b = copy.deepcopy(c['vector']), copy.deepcopy(c['bm25'])
    got = fusion_ref.weighted_ref(v, b, c['vector_weight'], c['bm25_weight'], c['top_k'])
    assert got == c['expected']
    assert v == c['vector_after'] and b == c['bm25_after']       # same in-place normalized_score writes


@pytest.mark.parametrize('name', sorted(GOLD['rerank']))
def test_rerank_matches_reference(name):
    c = GOLD['rerank'][name]
    seen = []

    def scorer(pairs):
        seen.extend(pairs)
        return c['scores'][:len(pairs)]
    got = rerank_ref.rerank_ref('the query', copy.deepcopy(c['results']), scorer, top_n=c['top_n'], top_k=c['top_k'])
    assert got == c['expected']
    if c['pairs'] is not None:
        assert [list(p) for p in seen] == [list(p) for p in c['pairs']]


@pytest.mark.parametrize('case', GOLD['cosine'], ids=lambda c: f"{c['dtype']}{len(c['a'])}")
def test_cosine_matches_reference(case):
    if case['dtype'] == 'float32':
        if np.__version__.split('.')[0] != GOLD['numpy_version'].split('.')[0]:
            pytest.skip('float32 scalar promotion differs between numpy major versions')
        a, b = np.asarray(case['a'], np.float32), np.asarray(case['b'], np.float32)
    else:
        a, b = case['a'], case['b']
    assert float(rerank_ref.cosine_ref(a, b)) == case['expected']


def test_stored_scores_are_rrf_sums():
    """Every fused score stored by the reference's evaluation runs is 1/(60+r) or a sum of two
    such terms (k = 60, ranks from 1, fp64); the few larger ones need three or more terms,
    i.e. a chunk_id occurring several times inside one list."""
    one = {1 / (60 + r) for r in range(1, 201)}
    two = {1 / (60 + a) + 1 / (60 + b) for a in range(1, 201) for b in range(1, 201)}
    n = big = 0
    for vals in STORED.values():
        for s in vals:
            n += 1
            if s in one or s in two:
                continue
            assert s > 2 / 61
            big += 1
    assert n == 90 and big <= 6
    # and the oracle reproduces one of them through its own code path
    ids_v = [f'v{i}' for i in range(100)]
    ids_b = [f'b{i}' for i in range(100)]
    ids_v[0] = ids_b[3] = 'x'
    assert dict((p, s) for p, s in fusion_ref.rrf_keys(ids_v, ids_b, top_k=1))[0] == 0.032018442622950824


WORDS = "aspirin ibuprofen pain fever dose tablet daily mg side effect nausea the of and for with".split()


def _docs(n, seed):
    rng = np.random.default_rng(seed)
    docs = [[WORDS[int(i)] for i in rng.integers(0, len(WORDS), int(rng.integers(0, 25)))] for _ in range(n)]
    for i in range(n):
        if i % 5:
            docs[i].append('common')
    return docs


@pytest.mark.parametrize('n,seed', [(7, 0), (60, 1), (400, 2)])
def test_bm25_array_form_is_bit_identical_to_literal_form(n, seed):
    docs = _docs(n, seed)
    lit = bm25_ref.BM25OkapiLiteral(docs)
    arr = bm25_ref.BM25OkapiArrays(docs)
    assert lit.avgdl == arr.avgdl and lit.average_idf == arr.average_idf
    for w, t in arr.vocab.items():
        assert lit.idf[w] == arr.idf[t]
    assert lit.idf['common'] == 0.25 * lit.average_idf
    for q in (['aspirin'], ['the', 'common', 'the'], ['zzz'], [], ['dose', 'zzz', 'mg', 'mg', 'common']):
        assert np.array_equal(lit.get_scores(q), arr.get_scores(q))


def test_bm25_hand_checked_values():
    docs = [['a', 'b'], ['a'], ['c', 'c', 'c', 'a']]
    lit = bm25_ref.BM25OkapiLiteral(docs, k1=1.5, b=0.75)
    import math
    idf_a = math.log(3 - 3 + 0.5) - math.log(3 + 0.5)           # negative -> floored
    idf_b = math.log(3 - 1 + 0.5) - math.log(1 + 0.5)
    idf_c = idf_b
    avg = (idf_a + idf_b + idf_c) / 3
    assert lit.idf['a'] == 0.25 * avg and lit.idf['b'] == idf_b
    avgdl = 7 / 3
    s = lit.get_scores(['c'])
    want = idf_c * (3 * 2.5 / (3 + 1.5 * (1 - 0.75 + 0.75 * 4 / avgdl)))
    assert s[0] == 0 and s[1] == 0 and s[2] == want
    assert list(bm25_ref.rank_rows_ref(s, 3)) == [2, 1, 0]        # ties: row descending


def test_bm25_search_window_and_filters():
    chunks = [{'chunk_id': f'c{i}', 'text': 't', 'source': 'a' if i % 2 else 'b'} for i in range(10)]
    scores = np.array([0, 5, 4, 0, 3, 2, 0, 1, 0, 0], float)
    got = bm25_ref.bm25_search_ref(scores, chunks, top_k=3)
    assert [g['chunk_id'] for g in got] == ['c1', 'c2', 'c4']
    got = bm25_ref.bm25_search_ref(scores, chunks, top_k=3, filters={'source': 'b'})
    assert [g['chunk_id'] for g in got] == ['c2', 'c4']           # window of 6, only the 'b' ones, fewer than 3
    got = bm25_ref.bm25_search_ref(np.zeros(10), chunks, top_k=2)
    assert [g['chunk_id'] for g in got] == ['c9', 'c8'] and got[0]['doc_id'] == '' and got[0]['metadata'] == {}


def test_dense_oracle_semantics():
    x = dense_ref.normalize_rows_ref(synth.dense_corpus(500, 64, seed=1) * 3.0)
    assert np.allclose(np.linalg.norm(x, axis=1), 1, atol=1e-6)
    q = x[17] * 5
    rows, scores = dense_ref.dense_topk_ref(x, q, 5)
    assert rows[0] == 17 and abs(scores[0] - 1) < 1e-6 and np.all(np.diff(scores) <= 0)
    mask = np.ones(500, np.uint8)
    mask[17] = 0
    rows2, _ = dense_ref.dense_topk_ref(x, q, 5, row_mask=mask)
    assert 17 not in rows2 and list(rows2[:4]) == list(rows[1:5])
    rows3, s3 = dense_ref.dense_topk_ref(x, q, 5, score_threshold=float(scores[2]))
    assert list(rows3) == list(rows[:3])
    # equal scores: row descending
    y = np.repeat(x[:1], 4, axis=0)
    assert list(dense_ref.dense_topk_ref(y, x[0], 3)[0]) == [3, 2, 1]
    rows_e, d = dense_ref.dense_topk_ref(x, x[3], 3, metric='euclidean')
    assert rows_e[0] == 3 and d[0] == 0 and np.all(np.diff(d) >= 0)
