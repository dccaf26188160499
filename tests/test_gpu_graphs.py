"""GPU: the encoder forwards of very small batches replayed from HIP graphs (BatchSearcher, RAG_GRAPHS) give the
answers of the eager path: same embeddings, same fused lists, same logits, same final rows, for several batches
in a row through the same captured graphs, with queries of different lengths."""
import numpy as np
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu


def _world(dev):
    import rag_hip
    from indexing import bert
    from indexing.bm25_indexer import build_arrays
    n, dim, vocab = 30000, 768, 2000
    corpus = synth.dense_corpus(n, dim, seed=51)
    term, doc, ln = synth.sparse_corpus(n, vocab, seed=52, mean_len=25)
    arr = build_arrays(term, doc, ln, vocab, 1.5, 0.75)
    dense = rag_hip.DenseIndex(corpus)
    bm25 = rag_hip.Bm25Index(arr.indptr, arr.rows, arr.impacts(), arr.idf, n)
    bi, _, _ = bert.load_model('random:bert-base', 'bi', dev, native=True, split_gemm=True)
    ce, _, _ = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.float32, native=True, split_gemm=True,
                               split_dtype=torch.float16)
    rng = np.random.default_rng(53)
    clen = ln + 2
    off = np.concatenate([[0], np.cumsum(clen)]).astype(np.int64)
    tok = rng.integers(1000, 30522, size=int(off[-1])).astype(np.int32)
    return dict(dense=dense, bm25=bm25, bi=bi, ce=ce, tok=torch.from_numpy(tok).to(dev), off=torch.from_numpy(off).to(dev),
                term=term, doc=doc, ln=ln, vocab=vocab)


def _queries(w, b, seed, dev):
    import rag_hip
    rng = np.random.default_rng(seed)
    qlen = rng.integers(5, 40, size=b)
    smax = int(qlen.max()) + 2
    ids = np.zeros((b, smax), np.int64)
    mask = np.zeros((b, smax), np.int64)
    ce_tok, ce_off = [], [0]
    for i in range(b):
        body = rng.integers(1000, 30522, size=int(qlen[i]))
        ids[i, 0], ids[i, 1:1 + qlen[i]], ids[i, 1 + qlen[i]] = 101, body, 102
        mask[i, :qlen[i] + 2] = 1
        ce_tok.append(body)
        ce_off.append(ce_off[-1] + int(qlen[i]))
    qterms, qoff = rag_hip.Bm25Index.flatten(synth.sparse_queries(w['term'], w['doc'], w['ln'], w['vocab'], b, seed=seed + 1))
    t = lambda a: torch.from_numpy(a).to(dev)
    return (t(ids), t(mask), t(qterms), t(qoff), int(qoff[-1]), t(np.concatenate(ce_tok).astype(np.int32)),
            t(np.asarray(ce_off, np.int64)))


@pytest.mark.parametrize('b', [1, 2])
def test_graph_replay_equals_the_eager_path(b, monkeypatch):
    from indexing.batch import BatchSearcher, TokenStore, pack_static
    import rag_hip
    rag_hip.init(0)
    dev = torch.device('cuda:0')
    w = _world(dev)
    mk = lambda: BatchSearcher(w['dense'], w['bm25'], bi_encoder=w['bi'], cross_encoder=w['ce'],
                               chunk_tokens=TokenStore(w['tok'], w['off'], 0), top_k=5, rerank_top_n=20)
    graphed = mk()
    monkeypatch.setenv('RAG_GRAPHS', '0')
    eager = mk()
    monkeypatch.delenv('RAG_GRAPHS')
    assert graphed.graph_max_batch >= 2 and graphed.graph_max_batch_ce >= 2 and eager.graph_max_batch == 0
    for seed in (1, 2, 3, 4):
        q = _queries(w, b, 100 * b + seed, dev)
        a = graphed.search(*q)
        e = eager.search(*q)
        torch.cuda.synchronize()
        assert torch.equal(a['fused_row'], e['fused_row']) and torch.equal(a['fused_score'], e['fused_score'])
        assert float((a['logits'] - e['logits']).abs().max()) < 2e-6        # same kernels on the same rows of a longer stream
        assert torch.equal(a['rows'], e['rows']) and torch.equal(a['count'], e['count'])
    # both forwards really came from captured graphs, one per shape
    kinds = sorted(k[0] for k, g in graphed._graphs.graphs.items() if g is not None)
    assert 'embed' in kinds and 'rerank' in kinds
    assert not eager._graphs.graphs
    # the static packing is the boolean-mask packing
    ids, mask = q[0], q[1]
    p_ids, p_types, p_pos, cu, longest = w['bi'].bert.pack(ids, mask)
    s_ids, s_types, s_pos, s_cu = pack_static(ids, mask)
    n = int(cu[-1])
    assert torch.equal(cu, s_cu) and torch.equal(p_ids, s_ids[:n]) and torch.equal(p_pos, s_pos[:n])
    assert int(s_ids[n:].abs().sum()) == 0
    w['dense'].close()
    w['bm25'].close()


def test_graphed_rerank_with_empty_trailing_slots_on_a_full_token_grid(monkeypatch):
    """A pool shorter than rerank_top_n leaves trailing slots without tokens (fused row -1); when the live pairs fill the
    2048-token grid exactly, those slots start AT the end of the stream.  The captured forward must still have a row
    there to read: same logits as the eager route, no device-side fault."""
    from indexing import bert
    from indexing.batch import BatchSearcher, TokenStore
    import rag_hip
    rag_hip.init(0)
    dev = torch.device('cuda:0')
    ce, _, _ = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.float32, native=True, split_gemm=True,
                               split_dtype=torch.float16)
    rng = np.random.default_rng(61)
    q_len, t = 10, 20
    clen = np.asarray([107] * 16 + [115] + [40] * 8)                      # pairs of 120 x 16 + 128 = 2048 tokens
    off = np.concatenate([[0], np.cumsum(clen)]).astype(np.int64)
    tok = torch.from_numpy(rng.integers(1000, 30522, size=int(off[-1])).astype(np.int32)).to(dev)
    store = TokenStore(tok, torch.from_numpy(off).to(dev), 0)
    q_tok = torch.from_numpy(rng.integers(1000, 30522, size=q_len).astype(np.int32)).to(dev)
    q_off = torch.tensor([0, q_len], dtype=torch.int64, device=dev)
    fused_row = torch.full((1, t), -1, dtype=torch.int64, device=dev)
    fused_row[0, :17] = torch.arange(17, device=dev)
    fused_cnt = torch.tensor([17], dtype=torch.int32, device=dev)
    mk = lambda: BatchSearcher(None, None, cross_encoder=ce, chunk_tokens=store, top_k=5, rerank_top_n=t)
    graphed = mk()
    monkeypatch.setenv('RAG_GRAPHS', '0')
    eager = mk()
    monkeypatch.delenv('RAG_GRAPHS')
    plen = rag_hip.ce_pair_len_dev(q_off, store.off, 0, torch.zeros(t, dtype=torch.int32, device=dev),
                                   fused_row.reshape(-1), 512)
    assert int(plen.sum()) == 2048 and int((plen == 0).sum()) == 3
    for _ in range(3):                                                   # capture, then replays
        ia, ca, la = graphed.rerank(q_tok, q_off, fused_row, fused_cnt)
        ie, ce_, le = eager.rerank(q_tok, q_off, fused_row, fused_cnt)
        torch.cuda.synchronize()
        assert torch.equal(ia, ie) and torch.equal(ca, ce_)
        assert float((la[0, :17] - le[0, :17]).abs().max()) < 2e-6
    assert any(k[0] == 'rerank' and g is not None for k, g in graphed._graphs.graphs.items())
    rag_hip.check_faults()


def test_profile_hooks_around_a_graphed_search():
    """The three measurement hooks record HIP event pairs around their launches; on a capturing stream they must record
    nothing (an event recorded into a graph cannot be read back).  With every hook switched on, a single query is
    captured and replayed, the hooks' reads return, and the results equal those with the hooks off."""
    from indexing.batch import BatchSearcher, TokenStore
    import rag_hip
    rag_hip.init(0)
    dev = torch.device('cuda:0')
    w = _world(dev)
    mk = lambda: BatchSearcher(w['dense'], w['bm25'], bi_encoder=w['bi'], cross_encoder=w['ce'],
                               chunk_tokens=TokenStore(w['tok'], w['off'], 0), top_k=5, rerank_top_n=20)
    q = _queries(w, 1, 4242, dev)
    want = mk().search(*q)
    torch.cuda.synchronize()
    w['dense'].profile(True)
    rag_hip.enc_mlp_profile(True)
    rag_hip.enc_linear_ln_split_profile(True)
    try:
        s = mk()
        outs = [s.search(*q) for _ in range(3)]            # capture (hooks on), then two replays
        torch.cuda.synchronize()
        assert any(k[0] == 'rerank' and g is not None for k, g in s._graphs.graphs.items())
        scan_ms = w['dense'].profile_read()                # eager launches (the dense scan is not in a graph) were timed ...
        assert len(scan_ms) == 3 and float(scan_ms.min()) > 0
        ms, rows = rag_hip.enc_mlp_profile_read()          # ... and the reads of the encoder hooks return (no launch of theirs
        ms2, rows2, k2 = rag_hip.enc_linear_ln_split_profile_read()   # outside a capture, so nothing to report or everything finite)
        assert np.all(np.isfinite(ms)) and np.all(np.isfinite(ms2))
    finally:
        w['dense'].profile(False)
        rag_hip.enc_mlp_profile(False)
        rag_hip.enc_linear_ln_split_profile(False)
    for o in outs:
        assert torch.equal(o['rows'], want['rows']) and float((o['logits'] - want['logits']).abs().max()) < 2e-6
    rag_hip.check_faults()
    w['dense'].close()
    w['bm25'].close()


def test_the_graph_cache_is_bounded(monkeypatch):
    """A serving loop with varied shapes must not keep one captured forward (and its memory pool) per shape for ever."""
    from indexing.bert import GraphCache
    made = []
    monkeypatch.setattr('indexing.bert.Graphed', lambda fn, inputs: made.append(1) or (lambda *a: None))
    c = GraphCache(limit=4)
    for i in range(10):
        c.get(('k', i), None, [])
    assert len(c.graphs) == 4 and ('k', 9) in c.graphs and ('k', 0) not in c.graphs and len(made) == 10


def test_class_api_single_calls_replay_graphs(monkeypatch):
    """MedicalEmbedder.encode of one text and CrossEncoderReranker.score_pairs of one pool (what the reference's
    query script calls per query) replay captured graphs: same embeddings and logits as the eager calls."""
    from indexing import MedicalEmbedder
    from reranker import CrossEncoderReranker
    emb = MedicalEmbedder(model_name='random:bert-base', device='cuda')
    ce = CrossEncoderReranker(model_name='random:minilm-l6', top_n=50, device='cuda')
    rng = np.random.default_rng(7)
    words = ['w%04d' % i for i in range(3000)]
    text = lambda n: ' '.join(words[j] for j in rng.integers(0, len(words), size=n))
    queries = [text(int(rng.integers(3, 20))) for _ in range(6)]
    pools = [[text(int(rng.integers(10, 90))) for _ in range(50)] for _ in range(3)] + [[text(30) for _ in range(7)]]
    got_e = [emb.encode(q) for q in queries] + [emb.encode(queries[0])]
    got_l = [ce.score_pairs([(queries[i], t) for t in pool]) for i, pool in enumerate(pools)]
    assert any(g is not None for g in emb._graphs.graphs.values())
    assert any(g is not None for g in ce._graphs.graphs.values())
    monkeypatch.setenv('RAG_GRAPHS', '0')
    emb._graphs.graphs.clear()
    ce._graphs.graphs.clear()
    want_e = [emb.encode(q) for q in queries] + [emb.encode(queries[0])]
    want_l = [ce.score_pairs([(queries[i], t) for t in pool]) for i, pool in enumerate(pools)]
    assert not emb._graphs.graphs and not ce._graphs.graphs
    for a, b in zip(got_e, want_e):
        assert a.shape == b.shape == (1, 768) and float(np.abs(a - b).max()) < 1e-6
    for a, b in zip(got_l, want_l):
        assert len(a) == len(b) and float(np.abs(np.asarray(a) - np.asarray(b)).max()) < 2e-6


def test_bm25_on_a_second_stream_changes_nothing(monkeypatch):
    """Batches of four queries and more run the BM25 search on a second stream beside the embedder and the dense scan
    (RAG_OVERLAP_BM25=0 keeps one stream): identical results, several batches in a row."""
    from indexing.batch import BatchSearcher, TokenStore
    import rag_hip
    rag_hip.init(0)
    dev = torch.device('cuda:0')
    w = _world(dev)
    mk = lambda: BatchSearcher(w['dense'], w['bm25'], bi_encoder=w['bi'], cross_encoder=w['ce'],
                               chunk_tokens=TokenStore(w['tok'], w['off'], 0), top_k=5, rerank_top_n=20)
    two = mk()
    monkeypatch.setenv('RAG_OVERLAP_BM25', '0')
    one = mk()
    monkeypatch.delenv('RAG_OVERLAP_BM25')
    assert two.overlap_bm25 and not one.overlap_bm25
    for seed in (1, 2, 3):
        q = _queries(w, 37, 900 + seed, dev)
        a, e = two.search(*q), one.search(*q)
        torch.cuda.synchronize()
        for key in ('fused_row', 'fused_score', 'fused_count', 'rows', 'count', 'logits'):
            assert torch.equal(a[key], e[key]), key
    assert two._bm25_stream is not None and one._bm25_stream is None
    w['dense'].close()
    w['bm25'].close()
