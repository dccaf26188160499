"""Oracle: rank fusion (test infrastructure, never imported by the product).

Restates ``HybridIndexer._reciprocal_rank_fusion`` and ``_weighted_fusion``
(reference ``src/indexing/hybrid_indexer.py:112-156`` and ``:158-218``) and the
dispatch in ``HybridIndexer.search`` (``:60-110``).

Pinned by ``tests/golden/reference_pure_python.json`` (outputs of the reference
functions themselves) and by ``tests/golden/results_rrf_scores.json`` (fused scores
stored in the reference's ``results/*.json``, which decode exactly as sums of
``1/(60+rank)``).

Both functions work on lists of result dicts like the reference.  The ``*_keys``
variants work on bare id / score sequences and return positions into the
concatenation ``vector_results + bm25_results``; they are what the device kernels
are compared with.
"""


def rrf_keys(vec_ids, bm_ids, top_k=10, k=60):
    """-> list of (position of the payload-supplying entry, fused score)."""
    score = {}
    first_pos = {}
    pos = 0
    for ids in (vec_ids, bm_ids):
        for rank, cid in enumerate(ids, 1):
            score[cid] = score.get(cid, 0) + 1 / (k + rank)
            if cid not in first_pos:
                first_pos[cid] = pos
            pos += 1
    ranked = sorted(score.items(), key=lambda kv: kv[1], reverse=True)   # stable
    return [(first_pos[cid], float(s)) for cid, s in ranked[:top_k]]


def rrf_ref(vector_results, bm25_results, top_k=10, k=60):
    both = list(vector_results) + list(bm25_results)
    picks = rrf_keys([r['chunk_id'] for r in vector_results],
                     [r['chunk_id'] for r in bm25_results], top_k=top_k, k=k)
    out = []
    for pos, s in picks:
        r = both[pos].copy()
        r['score'] = s
        r['fusion_method'] = 'rrf'
        out.append(r)
    return out


def _minmax(scores):
    """Per-list min-max normalisation; None when the reference leaves the list alone
    (empty list, or all scores equal)."""
    if not scores:
        return None
    lo, hi = min(scores), max(scores)
    if hi == lo:
        return None
    return [(s - lo) / (hi - lo) for s in scores]


def weighted_keys(vec_ids, vec_scores, bm_ids, bm_scores, vector_weight=0.5, bm25_weight=0.5, top_k=10):
    """-> (list of (payload position, fused score), vec_norm or None, bm_norm or None).

    Payload rule of the reference: a vector entry always (re)claims its id, a BM25
    entry only when the id is still unclaimed."""
    vn = _minmax(list(vec_scores))
    bn = _minmax(list(bm_scores))
    v_used = vn if vn is not None else list(vec_scores)
    b_used = bn if bn is not None else list(bm_scores)
    total = {}
    owner = {}
    for i, cid in enumerate(vec_ids):
        total[cid] = total.get(cid, 0) + vector_weight * v_used[i]
        owner[cid] = i
    nv = len(vec_ids)
    for i, cid in enumerate(bm_ids):
        total[cid] = total.get(cid, 0) + bm25_weight * b_used[i]
        if cid not in owner:
            owner[cid] = nv + i
    ranked = sorted(total.items(), key=lambda kv: kv[1], reverse=True)   # stable
    return [(owner[cid], float(s)) for cid, s in ranked[:top_k]], vn, bn


def weighted_ref(vector_results, bm25_results, vector_weight=0.5, bm25_weight=0.5, top_k=10):
    """Mutates its inputs like the reference: adds ``normalized_score`` to every entry
    of a list that gets normalised."""
    picks, vn, bn = weighted_keys(
        [r['chunk_id'] for r in vector_results], [r['score'] for r in vector_results],
        [r['chunk_id'] for r in bm25_results], [r['score'] for r in bm25_results],
        vector_weight, bm25_weight, top_k)
    if vn is not None:
        for r, v in zip(vector_results, vn):
            r['normalized_score'] = v
    if bn is not None:
        for r, v in zip(bm25_results, bn):
            r['normalized_score'] = v
    both = list(vector_results) + list(bm25_results)
    out = []
    for pos, s in picks:
        r = both[pos].copy()
        r['score'] = s
        r['fusion_method'] = 'weighted'
        out.append(r)
    return out


def hybrid_search_ref(vector_search, bm25_search, query, top_k=10, vector_weight=0.5,
                      bm25_weight=0.5, filters=None, fusion_method='rrf'):
    """Dispatch of ``HybridIndexer.search``: each retriever is asked for ``2*top_k``;
    a missing retriever contributes an empty list."""
    v = vector_search(query, top_k=top_k * 2, filters=filters) if vector_search else []
    b = bm25_search(query, top_k=top_k * 2, filters=filters) if bm25_search else []
    if fusion_method == 'rrf':
        return rrf_ref(v, b, top_k=top_k)
    return weighted_ref(v, b, vector_weight=vector_weight, bm25_weight=bm25_weight, top_k=top_k)
