"""GPU: the drop-in classes, the scripts and the batched device path end to end on the real
kernels, against the CPU oracle; plus shard-merge parity with 8 logical shards on one GPU."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from oracle import bm25_ref, fusion_ref, tokenize_ref
from oracle.dense_ref import dense_topk_ref, normalize_rows_ref
from tests import synth
from tests.test_host_cpu import make_chunks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_smoke_entry_point():
    import __graft_entry__
    __graft_entry__.smoke()


def _oracle_retrievers(chunks, emb, embedder):
    lit = bm25_ref.BM25OkapiLiteral([tokenize_ref.tokenize_ref(c['text']) for c in chunks])

    def vec(query, top_k, filters=None):
        mask = None
        if filters:
            mask = np.array([all(c.get(k) == v for k, v in filters.items()) for c in chunks], np.uint8)
        rows, scores = dense_topk_ref(emb, embedder.encode(query)[0], top_k, row_mask=mask)
        return [dict(chunk_id=chunks[r]['chunk_id'], doc_id=chunks[r]['doc_id'], source=chunks[r]['source'],
                     chunk_type=chunks[r]['chunk_type'], text=chunks[r]['text'], metadata=chunks[r]['metadata'],
                     score=float(s)) for r, s in zip(rows, scores)]

    def bm(query, top_k, filters=None):
        return bm25_ref.bm25_search_ref(lit.get_scores(tokenize_ref.tokenize_ref(query)), chunks, top_k=top_k,
                                        filters=filters)
    return vec, bm


def test_hybrid_indexer_matches_oracle_end_to_end(tmp_path):
    from indexing import MedicalEmbedder, VectorIndexer, BM25Indexer, HybridIndexer
    chunks = make_chunks(1500, seed=21)
    e = MedicalEmbedder(model_name='random:tiny', device='cuda')
    h = HybridIndexer(VectorIndexer('c', e, str(tmp_path / 'v')), BM25Indexer(), storage_path=str(tmp_path / 'i'))
    h.index_chunks(chunks)
    emb = normalize_rows_ref(e.encode([c['text'] for c in chunks]))
    vec, bm = _oracle_retrievers(chunks, emb, e)
    for query in ('aspirin pain dose', 'liver warning for children', 'nothing zzz', 'the heart blood pressure'):
        for method in ('rrf', 'weighted'):
            for filt in (None, {'source': 'pubmed'}):
                want = fusion_ref.hybrid_search_ref(vec, bm, query, top_k=10, vector_weight=0.7, bm25_weight=0.3,
                                                    filters=filt, fusion_method=method)
                got = h.search(query, top_k=10, vector_weight=0.7, bm25_weight=0.3, filters=filt,
                               fusion_method=method)
                assert [g['chunk_id'] for g in got] == [w['chunk_id'] for w in want], (query, method, filt)
                if method == 'rrf':
                    assert [g['score'] for g in got] == [w['score'] for w in want]       # integer ranks: exact
                else:
                    assert np.allclose([g['score'] for g in got], [w['score'] for w in want], atol=1e-4)


def test_scripts_1k_chunks_bm25_only(tmp_path):
    """The smallest benchmark configuration on the real kernels: build with scripts/build_index.py, query
    BM25-only top_k=5, ids and scores identical to the oracle."""
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import build_index
    import query as query_script
    chunks = make_chunks(1000, seed=11)
    cp = tmp_path / 'chunks.jsonl'
    with open(cp, 'w') as f:
        for c in chunks:
            f.write(json.dumps(c) + '\n')
    out_dir = tmp_path / 'indices'
    build_index.main(build_index.build_parser().parse_args([
        '--chunks_path', str(cp), '--embedding_model', 'random:tiny', '--output_dir', str(out_dir),
        '--vector_storage_path', str(tmp_path / 'vdb'), '--drug_mapping_path', str(tmp_path / 'none.json')]))
    meta = json.load(open(out_dir / 'index_metadata.json'))
    hybrid = query_script.load_hybrid_indexer(str(out_dir), meta)
    lit = bm25_ref.BM25OkapiLiteral([tokenize_ref.tokenize_ref(c['text']) for c in chunks])
    for q in ('What are the side effects of aspirin?', 'kidney liver', 'unknownword'):
        want = bm25_ref.bm25_search_ref(lit.get_scores(tokenize_ref.tokenize_ref(q)), chunks, top_k=5)
        assert hybrid.bm25_indexer.search(q, top_k=5) == want
    a = query_script.build_parser().parse_args(['--query', 'aspirin pain', '--indices_dir', str(out_dir),
                                                '--reranker_kind', 'simple', '--rerank_top_n', '15'])
    a.filters = None
    out = query_script.main(a)
    assert out['num_results'] == 5 and all('rerank_score' in r for r in out['results'])


def test_eight_logical_shards_merge_to_the_unsharded_lists():
    import rag_hip
    from indexing.bm25_indexer import build_arrays
    dev = torch.device('cuda:0')
    n, dim, vocab, b, k = 40000, 768, 3000, 48, 100
    corpus = synth.dense_corpus(n, dim, seed=31)
    queries = synth.dense_queries(corpus, b, seed=32)
    term, doc, ln = synth.sparse_corpus(n, vocab, seed=33, mean_len=20)
    bm_q = synth.sparse_queries(term, doc, ln, vocab, b, seed=34)
    full_d = rag_hip.DenseIndex(corpus)
    arr = build_arrays(term, doc, ln, vocab, 1.5, 0.75)
    full_b = rag_hip.Bm25Index(arr.indptr, arr.rows, arr.impacts(), arr.idf, n)
    want_d = full_d.search(queries, k)
    want_b = full_b.search(bm_q, k)
    stats = dict(df=np.diff(arr.indptr), total_len=int(ln.sum()), n_docs=n)
    qd = torch.from_numpy(queries).to(dev)
    qt, qo = rag_hip.Bm25Index.flatten(bm_q)
    qt, qo = torch.from_numpy(qt).to(dev), torch.from_numpy(qo).to(dev)
    ds, dr, dc, bs, br, bc = [], [], [], [], [], []
    shards = 8
    for s in range(shards):
        lo, hi = n * s // shards, n * (s + 1) // shards
        sd = rag_hip.DenseIndex(corpus[lo:hi], row_base=lo)
        x = sd.search_dev(qd, k)
        ds.append(x[0]); dr.append(x[1]); dc.append(x[2])
        sel = (doc >= lo) & (doc < hi)
        a = build_arrays(term[sel], doc[sel] - lo, ln[lo:hi], vocab, 1.5, 0.75, global_stats=stats)
        assert np.array_equal(a.idf, arr.idf) and a.avgdl == arr.avgdl
        sb = rag_hip.Bm25Index(a.indptr, a.rows, a.impacts(), a.idf, hi - lo, row_base=lo)
        y = sb.search_dev(qt, qo, int(qo[-1]), k)
        bs.append(y[0]); br.append(y[1]); bc.append(y[2])
        torch.cuda.synchronize()
    got_d = rag_hip.merge_topk_dev(torch.stack(ds, 1).contiguous(), torch.stack(dr, 1).contiguous(),
                                   torch.stack(dc, 1).contiguous(), k)
    got_b = rag_hip.merge_topk_dev(torch.stack(bs, 1).contiguous(), torch.stack(br, 1).contiguous(),
                                   torch.stack(bc, 1).contiguous(), k)
    torch.cuda.synchronize()
    assert np.array_equal(got_d[1].cpu().numpy(), want_d[1])
    assert np.array_equal(got_d[0].cpu().numpy(), want_d[0])              # same fp32 chains per row
    assert np.array_equal(got_b[1].cpu().numpy(), want_b[1])
    assert np.array_equal(got_b[0].cpu().numpy(), want_b[0])              # fp64 bit for bit


def test_simple_reranker_scores_from_resident_rows():
    """rag_dense_rescore gives the cosine the simple reranker computes by re-embedding."""
    from indexing import MedicalEmbedder, VectorIndexer
    from reranker import SimpleSimilarityReranker
    chunks = make_chunks(300, seed=41)
    e = MedicalEmbedder(model_name='random:tiny', device='cuda')
    vi = VectorIndexer('c', e, use_in_memory=True)
    vi.create_collection()
    vi.index_chunks(chunks)
    rr = SimpleSimilarityReranker(embedder=e, top_n=20)
    q = 'aspirin dose for children'
    hits = vi.search(q, top_k=20)
    by_text = rr.score_pairs([(q, h['text']) for h in hits])
    score, row, cnt = vi.search_rows(e.encode(q), 20)
    by_rows = vi._index.rescore(vi.prepare_queries(e.encode(q)), row)[0]
    assert np.allclose(by_text, by_rows, atol=1e-5)


def test_simple_reranker_class_uses_resident_rows():
    from indexing import MedicalEmbedder, VectorIndexer
    from reranker import SimpleSimilarityReranker, build_reranker
    chunks = make_chunks(400, seed=43)
    e = MedicalEmbedder(model_name='random:tiny', device='cuda')
    vi = VectorIndexer('c', e, use_in_memory=True)
    vi.create_collection()
    vi.index_chunks(chunks)
    plain = SimpleSimilarityReranker(embedder=e, top_n=50)
    fast = build_reranker('simple', top_n=50, embedder=e, vector_indexer=vi)
    q = 'liver warning adults'
    results = vi.search(q, top_k=40) + [{'chunk_id': 'x', 'text': 'unseen text about kidney', 'score': 0.0}]
    a = plain.score_pairs([(q, r['text']) for r in results])
    b = fast.score_pairs([(q, r['text']) for r in results])
    assert np.allclose(a, b, atol=2e-6)
    got = [r['chunk_id'] for r in fast.rerank(q, [dict(r) for r in results], top_k=10)]
    want = [r['chunk_id'] for r in plain.rerank(q, [dict(r) for r in results], top_k=10)]
    assert got == want


def test_batched_simple_rerank_matches_the_oracle():
    """BatchSearcher(simple_rerank=True): hybrid + RRF, then the reference's simple reranker (cosine of the
    query embedding with every pool entry's embedding, rerank.py:46-68) from the resident rows."""
    import rag_hip
    from indexing import bert
    from indexing.batch import BatchSearcher
    from indexing.bm25_indexer import build_arrays
    from oracle.rerank_ref import rerank_order_ref
    dev = torch.device('cuda:0')
    n, dim, vocab, b, top_n = 30000, 64, 2000, 40, 30
    corpus = synth.dense_corpus(n, dim, seed=5)
    term, doc, ln = synth.sparse_corpus(n, vocab, seed=6, mean_len=25)
    arr = build_arrays(term, doc, ln, vocab, 1.5, 0.75)
    dense = rag_hip.DenseIndex(corpus)
    bm25 = rag_hip.Bm25Index(arr.indptr, arr.rows, arr.impacts(), arr.idf, n)
    bi, _, _ = bert.load_model('random:tiny', 'bi', dev)
    rng = np.random.default_rng(8)
    q_ids = np.zeros((b, 12), np.int64)
    q_ids[:, 0], q_ids[:, 11] = 101, 102
    q_ids[:, 1:11] = rng.integers(1000, 4096, size=(b, 10))
    qterms, qoff = rag_hip.Bm25Index.flatten(synth.sparse_queries(term, doc, ln, vocab, b, seed=9))
    t = lambda a: torch.from_numpy(a).to(dev)
    s = BatchSearcher(dense, bm25, canon=None, bi_encoder=bi, top_k=5, rerank_top_n=top_n, simple_rerank=True)
    assert s.fetch_k == top_n
    out = s.search(t(q_ids), t(np.ones((b, 12), np.int64)), t(qterms), t(qoff), int(qoff[-1]))
    torch.cuda.synchronize()
    qv = s.embed(t(q_ids), t(np.ones((b, 12), np.int64))).cpu().numpy()
    fused = out['fused_row'].cpu().numpy()
    for i in range(b):
        cos = (corpus[fused[i]] @ qv[i]).astype(np.float32)
        got = out['logits'][i].cpu().numpy()
        assert np.abs(got - cos).max() < 1e-6
        order = rerank_order_ref([float(x) for x in got], 5)
        assert out['rows'][i].cpu().tolist() == [int(fused[i][p]) for p in order]
    with pytest.raises(ValueError):
        BatchSearcher(rag_hip.DenseIndex(corpus[:100], metric='dot'), None, bi_encoder=bi, simple_rerank=True)


def test_pipelined_batches_equal_one_by_one():
    """search_many (retrieval of batch i+1 on a side stream under the rerank of batch i) returns
    bit for bit what search returns batch by batch."""
    import rag_hip
    from indexing import bert
    from indexing.batch import BatchSearcher, TokenStore
    from indexing.bm25_indexer import build_arrays
    from tests import synth
    dev = torch.device('cuda:0')
    n, dim, vocab, b = 20000, 64, 3000, 64
    corpus = synth.dense_corpus(n, dim, seed=1)
    term, doc, ln = synth.sparse_corpus(n, vocab, seed=2, mean_len=25)
    arr = build_arrays(term, doc, ln, vocab, 1.5, 0.75)
    dense = rag_hip.DenseIndex(corpus)
    bm25 = rag_hip.Bm25Index(arr.indptr, arr.rows, arr.impacts(), arr.idf, n)
    bi, _, _ = bert.load_model('random:tiny', 'bi', dev)
    ce, _, _ = bert.load_model('random:tiny', 'cross', dev)
    rng = np.random.default_rng(3)
    off = np.concatenate([[0], np.cumsum(ln + 2)]).astype(np.int64)
    tok = rng.integers(1000, 4096, size=int(off[-1])).astype(np.int32)
    s = BatchSearcher(dense, bm25, canon=torch.from_numpy(np.arange(n, dtype=np.uint32)).to(dev), bi_encoder=bi,
                      cross_encoder=ce, chunk_tokens=TokenStore(torch.from_numpy(tok).to(dev), torch.from_numpy(off).to(dev)),
                      top_k=5, rerank_top_n=20)
    t = lambda a: torch.from_numpy(a).to(dev)
    batches = []
    for i in range(4):
        q_ids = np.zeros((b, 12), np.int64)
        q_ids[:, 0], q_ids[:, 11] = 101, 102
        q_ids[:, 1:11] = rng.integers(1000, 4096, size=(b, 10))
        qterms, qoff = rag_hip.Bm25Index.flatten(synth.sparse_queries(term, doc, ln, vocab, b, seed=10 + i))
        batches.append((t(q_ids), t(np.ones((b, 12), np.int64)), t(qterms), t(qoff), int(qoff[-1]),
                        t(q_ids[:, 1:11].astype(np.int32).reshape(-1)), t((np.arange(b + 1) * 10).astype(np.int64))))
    one_by_one = [s.search(*bt) for bt in batches]
    torch.cuda.synchronize()
    for _ in range(2):                                  # twice: the second pass reuses freed blocks of both streams
        piped = s.search_many(batches)
        torch.cuda.synchronize()
        for a, c in zip(one_by_one, piped):
            for key in ('rows', 'scores', 'rerank_score', 'count', 'fused_row', 'fused_score', 'fused_count'):
                assert torch.equal(a[key], c[key]), key
