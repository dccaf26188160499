"""Oracle: reranking (test infrastructure, never imported by the product).

Restates ``BaseReranker.rerank`` (reference ``src/reranker/rerank.py:29-43``) and the
pure-Python cosine of ``SimpleSimilarityReranker`` (``:53-68``).

Pinned by ``tests/golden/reference_pure_python.json`` (outputs of the reference
``BaseReranker.rerank`` itself, driven by a stub scorer).
"""
import math


def rerank_order_ref(scores, top_k):
    """Positions (into the pool) of the reranked output: stable sort by score
    descending, cut at ``top_k``."""
    order = sorted(range(len(scores)), key=lambda i: scores[i], reverse=True)
    return order[:top_k]


def rerank_ref(query, results, score_pairs, top_n=50, top_k=10):
    if not results:
        return results
    pool = results[:min(top_n, len(results))]
    pairs = [(query, c.get('text', '') or '') for c in pool]
    scores = [float(s) for s in score_pairs(pairs)]
    rescored = []
    for c, s in zip(pool, scores):
        r = c.copy()
        r['rerank_score'] = s
        r['fusion_method'] = (r.get('fusion_method') or 'pre') + '+rerank'
        rescored.append(r)
    return [rescored[i] for i in rerank_order_ref(scores, top_k)]


def cosine_ref(a, b):
    """Element-by-element cosine with the reference's zero-vector rule.  Like the
    reference it multiplies and sums the elements in whatever scalar type iteration
    yields (numpy float32 scalars for embedder output, so a sequential float32 sum)."""
    num = sum(x * y for x, y in zip(a, b))
    da = math.sqrt(sum(x * x for x in a))
    db = math.sqrt(sum(y * y for y in b))
    return 0.0 if (da == 0 or db == 0) else (num / (da * db))
