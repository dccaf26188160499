#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run in the build container only).

Runs the reference's own pure-Python functions (rank fusion, BM25 tokenizer,
BaseReranker.rerank, the simple reranker's cosine) from the read-only reference
checkout, with inert placeholder modules registered for the third-party packages
those files import at module level but that the exercised functions never touch
(sentence_transformers, rank_bm25, qdrant_client, spacy, scispacy).  Only inputs and
the outputs the reference produced for them are written; no reference source is
copied.  It also collects the fused scores stored in the reference's result files.

Usage:  python tests/golden/make_golden.py [/root/reference]
"""
import copy
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Unavailable:
    def __init__(self, *a, **k):
        raise RuntimeError('third-party class is not available in this container')


def import_reference():
    _placeholder('sentence_transformers', SentenceTransformer=_Unavailable, CrossEncoder=_Unavailable)
    _placeholder('rank_bm25', BM25Okapi=_Unavailable)
    _placeholder('qdrant_client', QdrantClient=_Unavailable)
    _placeholder('qdrant_client.models', Distance=types.SimpleNamespace(COSINE=0, EUCLID=1, DOT=2),
                 VectorParams=_Unavailable, PointStruct=_Unavailable, Filter=_Unavailable,
                 FieldCondition=_Unavailable, MatchValue=_Unavailable)
    _placeholder('spacy', load=_Unavailable)
    _placeholder('scispacy')
    os.chdir(REF)
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, 'src'))
    from src.indexing.hybrid_indexer import HybridIndexer
    from src.indexing.bm25_indexer import BM25Indexer
    from src.reranker.rerank import BaseReranker, SimpleSimilarityReranker
    return HybridIndexer, BM25Indexer, BaseReranker, SimpleSimilarityReranker


def res(cid, score, **extra):
    d = {'chunk_id': cid, 'doc_id': 'd_' + cid, 'source': 's', 'chunk_type': 't',
         'text': 'text of ' + cid, 'metadata': {}, 'score': score}
    d.update(extra)
    return d


FUSION_CASES = {
    'basic_overlap': ([('a', 0.9), ('b', 0.8), ('c', 0.7), ('d', 0.6)],
                      [('c', 12.5), ('e', 9.0), ('a', 4.25), ('f', 1.0)], 10),
    'cut_top2': ([('a', 0.9), ('b', 0.8), ('c', 0.7), ('d', 0.6)],
                 [('c', 12.5), ('e', 9.0), ('a', 4.25), ('f', 1.0)], 2),
    'empty_vector': ([], [('x', 3.0), ('y', 2.0), ('z', 1.0)], 5),
    'empty_bm25': ([('x', 0.3), ('y', 0.2), ('z', 0.1)], [], 5),
    'both_empty': ([], [], 5),
    'dup_inside_vector': ([('a', 0.9), ('b', 0.8), ('a', 0.7), ('c', 0.6), ('a', 0.5)],
                          [('b', 5.0), ('a', 4.0), ('d', 3.0)], 10),
    'dup_inside_bm25': ([('p', 0.5), ('q', 0.4)],
                        [('r', 9.0), ('r', 8.0), ('p', 7.0), ('r', 6.0), ('q', 5.0)], 10),
    'all_ties': ([('a', 0.5), ('b', 0.5), ('c', 0.5)], [('d', 1.0), ('e', 1.0), ('f', 1.0)], 10),
    'const_scores_one_list': ([('a', 0.5), ('b', 0.5), ('c', 0.5)],
                              [('c', 3.0), ('b', 2.0), ('g', 1.0)], 10),
    'single_entry_lists': ([('a', 0.7)], [('a', 2.0)], 3),
    'disjoint': ([('a', 0.9), ('b', 0.1)], [('c', 7.0), ('d', 0.0)], 10),
    'negative_scores': ([('a', -0.1), ('b', -0.4), ('c', -0.2)], [('b', -1.0), ('d', -3.0), ('a', -2.0)], 10),
    'long_lists': ([('v%d' % (i % 37), 1.0 - 0.01 * i) for i in range(40)],
                   [('v%d' % ((i * 7) % 53), 30.0 - 0.37 * i) for i in range(40)], 20),
}

TOKENIZE_CASES = [
    "Pt w/ MI & HTN: 5 mg of co-trimoxazole (T2DM) a I x HbA1c",
    "What are the side effects of aspirin?",
    "",
    "a b c d",
    "w/o complications; Rx: 10mg q.d. x 7 d",
    "Café-au-lait spots — naïve T-cells, β-blockers & 5-HT3",
    "under_score __dunder__ 3.14 1,000 x1 y",
    "HR 72 BP 120/80 SOB? no. URI/COPD; pt c/o Sx",
    "漢字 テスト mixed русский text",
    "tabs\tand\nnewlines\r\nhere  double  spaces",
]

NORMALIZE_CASES = [
    "Pt w/ MI & HTN on ACE inhibitor; hx of T2DM, HbA1c 8.1",
    "What are the side effects of Tylenol and advil for a heart attack?",
    "high blood pressure treatment w/o NSAID",
    "COPD vs emphysema: flu shot? GERD / acid reflux / heartburn",
    "mi amigo has dm and a Stroke",          # case-insensitive whole-word matches
    "MIxed tokens HTNx xMI DM2",             # no whole-word match
    "",
    "generalized anxiety disorder (GAD) and MDD; sugar disease",
]
NORMALIZE_DRUG_MAPPING = {"acetaminophen": ["Tylenol", "Panadol"], "ibuprofen": ["Advil", "Motrin"]}

RERANK_CASES = {
    # name: (results as (cid, fused score, extra), scores by position, top_n, top_k)
    'basic': ([('a', 0.03, {'fusion_method': 'rrf'}), ('b', 0.02, {'fusion_method': 'rrf'}),
               ('c', 0.01, {'fusion_method': 'rrf'})], [0.1, 0.9, 0.5], 50, 10),
    'pool_smaller_than_results': ([('a', 0.5, {'fusion_method': 'weighted'}), ('b', 0.4, {'fusion_method': 'weighted'}),
                                   ('c', 0.3, {'fusion_method': 'weighted'}), ('d', 0.2, {'fusion_method': 'weighted'})],
                                  [1.0, 3.0, 2.0, 9.0], 3, 2),
    'ties_keep_pool_order': ([('a', 0.5, {}), ('b', 0.4, {}), ('c', 0.3, {}), ('d', 0.2, {})],
                             [0.5, 0.7, 0.5, 0.7], 50, 10),
    'no_fusion_method_and_missing_text': ([('a', 0.5, {'text': None}), ('b', 0.4, {'fusion_method': ''}),
                                           ('c', 0.3, {'fusion_method': None})], [3.0, 2.0, 1.0], 50, 10),
    'empty': ([], [], 50, 10),
    'top_k_zero_pool': ([('a', 0.5, {})], [0.25], 0, 10),
}


def main():
    HybridIndexer, BM25Indexer, BaseReranker, SimpleSimilarityReranker = import_reference()
    out = {'generator': 'tests/golden/make_golden.py', 'rrf': {}, 'weighted': {}, 'tokenize': [],
           'rerank': {}, 'cosine': []}

    hy = HybridIndexer.__new__(HybridIndexer)          # fusion methods use no instance state
    for name, (v, b, top_k) in FUSION_CASES.items():
        vr = [res(c, s) for c, s in v]
        br = [res(c, s) for c, s in b]
        got = hy._reciprocal_rank_fusion(copy.deepcopy(vr), copy.deepcopy(br), top_k=top_k)
        out['rrf'][name] = {'vector': vr, 'bm25': br, 'top_k': top_k, 'k': 60, 'expected': got}
        for vw, bw in ((0.5, 0.5), (0.7, 0.3)):
            vin, bin_ = copy.deepcopy(vr), copy.deepcopy(br)
            got = hy._weighted_fusion(vin, bin_, vector_weight=vw, bm25_weight=bw, top_k=top_k)
            out['weighted']['%s_w%02d' % (name, int(vw * 10))] = {
                'vector': vr, 'bm25': br, 'top_k': top_k, 'vector_weight': vw, 'bm25_weight': bw,
                'expected': got, 'vector_after': vin, 'bm25_after': bin_}

    bm = BM25Indexer.__new__(BM25Indexer)
    from src.preprocessing.medical_term_normalizer import MedicalTermNormalizer
    bm.normalizer = MedicalTermNormalizer()
    out['abbreviation_keys'] = list(bm.normalizer.abbreviations.keys())
    for text in TOKENIZE_CASES:
        out['tokenize'].append({'text': text, 'expected': bm.tokenize(text)})

    # query rewriting in front of retrieval (scripts/query.py:122-136 calls normalize_text through the
    # query preprocessor): default tables, then with a brand -> generic mapping
    out['normalize'] = []
    for mapping in ({}, NORMALIZE_DRUG_MAPPING):
        nz = MedicalTermNormalizer()
        nz.drug_mapping = dict(mapping)
        for text in NORMALIZE_CASES:
            for expand in (True, False):
                out['normalize'].append({'text': text, 'drug_mapping': mapping, 'expand_abbreviations': expand,
                                         'expected': nz.normalize_text(text, expand_abbreviations=expand),
                                         'terms': nz.extract_medical_terms(text) if text else None})

    class TableReranker(BaseReranker):
        def __init__(self, table, top_n):
            super().__init__(top_n=top_n, name='table')
            self.table = table
            self.seen_pairs = None

        def score_pairs(self, pairs):
            pairs = list(pairs)
            self.seen_pairs = pairs
            return self.table[:len(pairs)]

    for name, (rs, scores, top_n, top_k) in RERANK_CASES.items():
        results = []
        for cid, s, extra in rs:
            results.append(res(cid, s, **extra))
        rr = TableReranker(scores, top_n)
        got = rr.rerank('the query', copy.deepcopy(results), top_k=top_k)
        out['rerank'][name] = {'results': results, 'scores': scores, 'top_n': top_n, 'top_k': top_k,
                               'expected': got, 'pairs': rr.seen_pairs}

    import numpy as np
    rng = np.random.default_rng(7)
    for n in (3, 16, 96):
        a = rng.standard_normal(n).astype(np.float32)
        b = rng.standard_normal(n).astype(np.float32)
        out['cosine'].append({'a': a.tolist(), 'b': b.tolist(), 'dtype': 'float32',
                              'expected': float(SimpleSimilarityReranker._cosine(a, b))})
        out['cosine'].append({'a': a.tolist(), 'b': b.tolist(), 'dtype': 'pyfloat',
                              'expected': float(SimpleSimilarityReranker._cosine(a.tolist(), b.tolist()))})
    out['cosine'].append({'a': [0.0, 0.0], 'b': [1.0, 2.0], 'dtype': 'pyfloat',
                          'expected': float(SimpleSimilarityReranker._cosine([0.0, 0.0], [1.0, 2.0]))})
    out['numpy_version'] = np.__version__

    with open(os.path.join(HERE, 'reference_pure_python.json'), 'w', encoding='utf-8') as f:
        json.dump(out, f, ensure_ascii=True, separators=(',', ':'))

    # fused scores stored by the reference's evaluation runs (data files, not code)
    stored = {}
    for fname in ('kaggle_rag_test_results.json', 'openfda_rag_test_results.json'):
        with open(os.path.join(REF, 'results', fname), 'r', encoding='utf-8') as f:
            data = json.load(f)
        vals = []

        def walk(x):
            if isinstance(x, dict):
                if ('chunk_id' in x or 'doc_id' in x) and isinstance(x.get('score'), float):
                    vals.append(x['score'])
                for v in x.values():
                    walk(v)
            elif isinstance(x, list):
                for v in x:
                    walk(v)
        walk(data)
        stored[fname] = vals
    with open(os.path.join(HERE, 'results_rrf_scores.json'), 'w', encoding='utf-8') as f:
        json.dump({'source': 'score fields of the reference results/*.json files', 'scores': stored}, f, indent=1)
    print('wrote', {k: len(v) for k, v in stored.items()})


if __name__ == '__main__':
    main()
