"""Oracle-backed stand-ins for the ``rag_hip`` entry points (TEST INFRASTRUCTURE ONLY).

The product has no CPU path.  To exercise the host-side logic (result formatting, filters,
persistence, script plumbing, the multi-rank exchange) in a container without a GPU, the CPU
tests swap these doubles in for the ctypes-bound entry points.  They follow the same calling
conventions and compute with the oracle; nothing here ships or is measured.
"""
import numpy as np
import torch

from oracle.bm25_ref import rank_rows_ref
from oracle.dense_ref import dense_topk_ref
from oracle.fusion_ref import rrf_keys, weighted_keys
from oracle.rerank_ref import rerank_order_ref

MAX_K = 1024
MAX_FUSE_LIST = 512
_METRIC = {'cosine': 'cosine', 'dot': 'dot', 'euclidean': 'euclidean', 'euclid': 'euclidean'}


class DenseIndex:
    def __init__(self, rows, metric='cosine', row_base=0, borrow=False):
        self.rows = rows.cpu().numpy() if hasattr(rows, 'cpu') else np.ascontiguousarray(rows, np.float32)
        self.metric = _METRIC[metric]
        self.n_rows, self.dim = self.rows.shape
        self.row_base = int(row_base)

    def close(self):
        pass

    def search(self, queries, k, row_mask=None):
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, self.dim)
        b = q.shape[0]
        score = np.zeros((b, k), np.float32)
        row = np.full((b, k), -1, np.int64)
        cnt = np.zeros(b, np.int32)
        for i in range(b):
            if self.n_rows == 0:
                continue
            # the store already holds normalised rows; the oracle normalises the query again
            # for cosine, which is the identity on unit vectors up to rounding
            r, s = dense_topk_ref(self.rows, q[i], k, metric=self.metric if self.metric != 'cosine' else 'dot',
                                  row_mask=row_mask)
            cnt[i] = len(r)
            row[i, :len(r)] = r + self.row_base
            score[i, :len(r)] = s
        return score, row, cnt

    def search_dev(self, queries, k, row_mask=None, out=None):
        s, r, c = self.search(queries.numpy(), k, None if row_mask is None else row_mask.numpy())
        return torch.from_numpy(s), torch.from_numpy(r), torch.from_numpy(c)

    def rescore_dev(self, queries, cand_rows):
        return torch.from_numpy(self.rescore(queries.numpy(), cand_rows.numpy()))

    def rescore(self, queries, cand_rows):
        q = np.ascontiguousarray(queries, np.float32).reshape(-1, self.dim)
        c = np.asarray(cand_rows, np.int64).reshape(q.shape[0], -1)
        out = np.zeros(c.shape, np.float32)
        for i in range(q.shape[0]):
            for j, r in enumerate(c[i]):
                if 0 <= r - self.row_base < self.n_rows:
                    out[i, j] = np.float32(self.rows[r - self.row_base] @ q[i])
        return out

    def reserve(self, max_queries, k):
        pass

    def profile(self, enable=True):
        pass

    def profile_read(self):
        return np.zeros(0, np.float32)


class Bm25Index:
    def __init__(self, indptr, rows, impact, idf, n_docs, row_base=0):
        self.indptr = np.asarray(indptr, np.int64)
        self.rows = np.asarray(rows, np.int64)
        self.impact = np.asarray(impact, np.float64)
        self.idf = np.asarray(idf, np.float64)
        self.n_docs, self.row_base = int(n_docs), int(row_base)
        self.vocab = self.indptr.shape[0] - 1

    def close(self):
        pass

    @staticmethod
    def flatten(queries):
        off = np.zeros(len(queries) + 1, np.int32)
        for i, q in enumerate(queries):
            off[i + 1] = off[i] + len(q)
        terms = np.zeros(int(off[-1]), np.int32)
        for i, q in enumerate(queries):
            terms[off[i]:off[i + 1]] = q
        return terms, off

    def scores(self, query_terms):
        s = np.zeros(self.n_docs)
        for t in query_terms:
            t = int(t)
            if t < 0:
                continue
            lo, hi = self.indptr[t], self.indptr[t + 1]
            contrib = np.zeros(self.n_docs)
            contrib[self.rows[lo:hi]] = self.idf[t] * self.impact[lo:hi]
            s += contrib
        return s

    def search(self, queries, k):
        b = len(queries)
        score = np.zeros((b, k), np.float64)
        row = np.full((b, k), -1, np.int64)
        cnt = np.zeros(b, np.int32)
        for i, q in enumerate(queries):
            s = self.scores(q)
            order = rank_rows_ref(s, k)
            cnt[i] = len(order)
            row[i, :len(order)] = order + self.row_base
            score[i, :len(order)] = s[order]
        return score, row, cnt

    def reserve(self, max_queries, max_terms_total, k):
        pass

    def search_dev(self, qterms, qoff, n_terms_total, k, out=None):
        t, o = qterms.numpy(), qoff.numpy()
        s, r, c = self.search([t[o[i]:o[i + 1]].tolist() for i in range(len(o) - 1)], k)
        return torch.from_numpy(s), torch.from_numpy(r), torch.from_numpy(c)


def _canon(rows, canon):
    return [int(canon[r]) if canon is not None else int(r) for r in rows]


def _emit(picks, both, top_k):
    pos = np.full(top_k, -1, np.int32)
    row = np.full(top_k, -1, np.int64)
    score = np.zeros(top_k, np.float64)
    for j, (p, s) in enumerate(picks):
        pos[j], row[j], score[j] = p, both[p], s
    return pos, row, score, len(picks)


def fuse_rrf(vec_rows, bm_rows, top_k, canon=None, k_const=60.0):
    k = int(k_const) if float(k_const).is_integer() else k_const
    outs = [_emit(rrf_keys(_canon(v, canon), _canon(b, canon), top_k=top_k, k=k), list(v) + list(b), top_k)
            for v, b in zip(vec_rows, bm_rows)]
    return (np.stack([o[0] for o in outs]), np.stack([o[1] for o in outs]), np.stack([o[2] for o in outs]),
            np.asarray([o[3] for o in outs], np.int32))


def fuse_weighted(vec_rows, vec_scores, bm_rows, bm_scores, top_k, vector_weight=0.5, bm25_weight=0.5, canon=None):
    outs = []
    for v, vs, b, bs in zip(vec_rows, vec_scores, bm_rows, bm_scores):
        picks, _, _ = weighted_keys(_canon(v, canon), [float(x) for x in vs], _canon(b, canon),
                                    [float(x) for x in bs], vector_weight, bm25_weight, top_k)
        outs.append(_emit(picks, list(v) + list(b), top_k))
    return (np.stack([o[0] for o in outs]), np.stack([o[1] for o in outs]), np.stack([o[2] for o in outs]),
            np.asarray([o[3] for o in outs], np.int32))


def _lists(row, cnt):
    return [row[i, :int(cnt[i])].tolist() for i in range(row.shape[0])]


def fuse_rrf_dev(vec_row, vec_cnt, bm_row, bm_cnt, top_k, canon=None, k_const=60.0):
    c = None if canon is None else canon.numpy()
    out = fuse_rrf(_lists(vec_row.numpy(), vec_cnt.numpy()), _lists(bm_row.numpy(), bm_cnt.numpy()), top_k, c, k_const)
    return tuple(torch.from_numpy(x) for x in out)


def fuse_weighted_dev(vec_row, vec_score, vec_cnt, bm_row, bm_score, bm_cnt, top_k, vector_weight=0.5,
                      bm25_weight=0.5, canon=None):
    c = None if canon is None else canon.numpy()
    vl, bl = _lists(vec_row.numpy(), vec_cnt.numpy()), _lists(bm_row.numpy(), bm_cnt.numpy())
    vs = [vec_score.numpy()[i, :len(v)].tolist() for i, v in enumerate(vl)]
    bs = [bm_score.numpy()[i, :len(b)].tolist() for i, b in enumerate(bl)]
    return tuple(torch.from_numpy(x) for x in fuse_weighted(vl, vs, bl, bs, top_k, vector_weight, bm25_weight, c))


def topk_segmented(scores, seg_off, top_k):
    s = np.asarray(scores, np.float32)
    o = np.asarray(seg_off, np.int32)
    b = o.shape[0] - 1
    idx = np.full((b, top_k), -1, np.int32)
    cnt = np.zeros(b, np.int32)
    for q in range(b):
        order = rerank_order_ref(s[o[q]:o[q + 1]].tolist(), top_k)
        idx[q, :len(order)] = order
        cnt[q] = len(order)
    return idx, cnt


def topk_segmented_dev(scores, seg_off, top_k):
    i, c = topk_segmented(scores.numpy(), seg_off.numpy(), top_k)
    return torch.from_numpy(i), torch.from_numpy(c)


def merge_topk_dev(score, row, count, k_out, smaller_is_better=False):
    s, r = score.numpy(), row.numpy()
    b, nl, kin = s.shape
    c = count.numpy() if count is not None else None
    os_ = np.zeros((b, k_out), s.dtype)
    or_ = np.full((b, k_out), -1, np.int64)
    oc = np.zeros(b, np.int32)
    for q in range(b):
        items = []
        for l in range(nl):
            n = int(c[q, l]) if c is not None else kin
            items += [(s[q, l, j], int(r[q, l, j])) for j in range(n) if r[q, l, j] >= 0]
        items.sort(key=lambda t: ((-t[0]) if smaller_is_better else t[0], t[1]), reverse=True)
        items = items[:k_out]
        oc[q] = len(items)
        for j, (sc, rw) in enumerate(items):
            os_[q, j], or_[q, j] = sc, rw
    return torch.from_numpy(os_), torch.from_numpy(or_), torch.from_numpy(oc)


def topk_stable_f64(scores, top_k):
    return rerank_order_ref(list(scores), min(int(top_k), MAX_K))


def ce_pair_len_dev(query_off, chunk_off, chunk_row_base, pair_query, pair_row, max_len):
    out = []
    for q, r in zip(pair_query.tolist(), pair_row.tolist()):
        if r < 0:
            out.append(0)
            continue
        qn = int(query_off[q + 1] - query_off[q])
        cn = int(chunk_off[r - chunk_row_base + 1] - chunk_off[r - chunk_row_base])
        qn = min(qn, max(max_len - 3, 0))
        cn = min(cn, max(max_len - 3 - qn, 0))
        out.append(qn + cn + 3)
    return torch.tensor(out, dtype=torch.int32)


def ce_pack_dev(query_tok, query_off, chunk_tok, chunk_off, chunk_row_base, pair_query, pair_row, pair_slot,
                n_slots, seq_len, cls_id=101, sep_id=102, pad_id=0):
    ids = torch.full((n_slots, seq_len), pad_id, dtype=torch.int64)
    types = torch.zeros((n_slots, seq_len), dtype=torch.int64)
    mask = torch.zeros((n_slots, seq_len), dtype=torch.int64)
    for p, (q, r) in enumerate(zip(pair_query.tolist(), pair_row.tolist())):
        slot = p if pair_slot is None else int(pair_slot[p])
        if r < 0:
            continue
        qt = query_tok[int(query_off[q]):int(query_off[q + 1])].tolist()[:max(seq_len - 3, 0)]
        ct = chunk_tok[int(chunk_off[r - chunk_row_base]):int(chunk_off[r - chunk_row_base + 1])].tolist()
        ct = ct[:max(seq_len - 3 - len(qt), 0)]
        seq = [cls_id] + qt + [sep_id] + ct + [sep_id]
        ids[slot, :len(seq)] = torch.tensor(seq)
        types[slot, len(qt) + 2:len(seq)] = 1
        mask[slot, :len(seq)] = 1
    return ids, types, mask


FILTER_ANY, FILTER_NONE, FILTER_MISSING = -1, -2, -3


class RowFilter:
    """Host double of rag_hip.RowFilter: the same mask from the same code columns."""
    MAX_FIELDS = 8

    def __init__(self, columns, device='cpu'):
        self.keys = list(columns)
        self.cols = {k: np.asarray(v, np.int32) for k, v in columns.items()}
        self.n_rows = len(next(iter(self.cols.values()))) if self.cols else 0

    def mask_dev(self, want):
        mask = np.ones(self.n_rows, np.uint8)
        for k, w in want.items():
            if w != FILTER_ANY:
                mask &= (self.cols[k] == w).astype(np.uint8)
        return torch.from_numpy(mask)


def check_faults():
    """The doubles index nothing with unchecked device data: nothing to report."""


DOUBLES = ('DenseIndex', 'Bm25Index', 'fuse_rrf', 'fuse_weighted', 'fuse_rrf_dev', 'fuse_weighted_dev',
           'topk_segmented', 'topk_segmented_dev', 'topk_stable_f64', 'merge_topk_dev', 'ce_pair_len_dev', 'ce_pack_dev',
           'RowFilter', 'check_faults')


class _Setter:
    """monkeypatch-shaped setter for processes that are not pytest (a rank of a launched script)."""

    @staticmethod
    def setattr(obj, name, value):
        setattr(obj, name, value)


def install(monkeypatch=None):
    """Replace the GPU entry points of the imported ``rag_hip`` module with the doubles (for good when no
    monkeypatch fixture is given)."""
    import rag_hip
    monkeypatch = monkeypatch or _Setter
    for name in DOUBLES:
        monkeypatch.setattr(rag_hip, name, globals()[name])
    # the device-side index build is replaced by the host counting it is tested against on the GPU
    from indexing import bm25_indexer

    def build_on_host(term_ids, doc_ids, doc_len, vocab, k1, b, global_stats=None, device=None):
        return bm25_indexer.build_arrays(term_ids, doc_ids, doc_len, vocab, k1, b, global_stats=global_stats)
    monkeypatch.setattr(bm25_indexer, 'build_arrays_device', build_on_host)
    return rag_hip
