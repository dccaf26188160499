"""Oracle: exact dense nearest-neighbour search (test infrastructure, never imported
by the product).

PARITY UNPINNED at the ``qdrant-client`` boundary: the package
(``qdrant-client>=1.7.0``, reference ``requirements.txt:6``) is neither in the
reference checkout nor installed, and the reference holds no test or stored
output for it.  This file restates the published behaviour of qdrant-client's
local (embedded) mode as used by ``VectorIndexer.index_chunks`` /
``VectorIndexer.search`` (reference ``src/indexing/vector_indexer.py:105-230``):

* ``Distance.COSINE``: vectors are L2-normalised in float32 when inserted, the
  query is normalised at search time, score = float32 dot product;
* ``Distance.DOT``: score = float32 dot product of the raw vectors;
* ``Distance.EUCLID``: score = float32 Euclidean distance, smaller is better;
* exact brute force over all points that pass the payload filter, ordered best
  first, cut at ``limit``; ``score_threshold`` keeps scores at least as good.

Tie rule (implementation defined upstream, fixed here and in the HIP path): equal
scores are ordered by row index descending, i.e. ``argsort(kind='stable')[::-1]``
(for EUCLID: distance ascending, then row index descending).
"""
import numpy as np


def normalize_rows_ref(x):
    """float32 row normalisation as applied on insert for COSINE; zero rows are kept."""
    x = np.asarray(x, dtype=np.float32)
    norm = np.linalg.norm(x, axis=-1, keepdims=True).astype(np.float32)
    norm = np.where(norm == 0, np.float32(1), norm)
    return (x / norm).astype(np.float32)


def dense_scores_ref(rows, query, metric='cosine'):
    rows = np.asarray(rows, dtype=np.float32)
    q = np.asarray(query, dtype=np.float32).reshape(-1)
    if metric == 'cosine':
        q = normalize_rows_ref(q[None, :])[0]
        return rows @ q
    if metric == 'dot':
        return rows @ q
    if metric == 'euclidean':
        return np.linalg.norm(rows - q[None, :], axis=1).astype(np.float32)
    raise ValueError(metric)


def dense_topk_ref(rows, query, top_k, metric='cosine', row_mask=None, score_threshold=None):
    """Returns (row indices, float32 scores), best first.  ``rows`` must already be
    normalised for cosine (``normalize_rows_ref``), as the store holds them."""
    s = dense_scores_ref(rows, query, metric)
    n = s.shape[0]
    if metric == 'euclidean':
        order = np.lexsort((-np.arange(n), s))          # distance asc, row desc
    else:
        order = np.argsort(s, kind='stable')[::-1]      # score desc, row desc
    keep = []
    for i in order:
        if row_mask is not None and not row_mask[i]:
            continue
        if score_threshold is not None:
            if metric == 'euclidean':
                if s[i] > score_threshold:
                    continue
            elif s[i] < score_threshold:
                continue
        keep.append(i)
        if len(keep) >= top_k:
            break
    keep = np.asarray(keep, dtype=np.int64)
    return keep, s[keep]
