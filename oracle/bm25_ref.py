"""Oracle: BM25 indexing and search (test infrastructure, never imported by the product).

PARITY UNPINNED at the ``rank_bm25`` boundary: the package (``rank_bm25>=0.2.1``,
reference ``requirements.txt:7``) is neither in the reference checkout nor
installed, and the reference holds no test or stored output for it.  This file
restates the published ``BM25Okapi`` algorithm of rank_bm25 0.2.x and the
reference's own call sites:

* ``BM25Indexer.index_chunks``  reference ``src/indexing/bm25_indexer.py:80-111``
* ``BM25Indexer.search``        reference ``src/indexing/bm25_indexer.py:113-163``

Two forms are kept on purpose:

``BM25OkapiLiteral``  one ``{term: count}`` dict per document and a Python pass
    over all documents per query token - the shape of the upstream code.
``BM25OkapiArrays``   the same arithmetic on numpy arrays (term-major postings);
    every floating-point operation is applied in the same order to the same
    operands, so the scores are bit-identical to the literal form (checked in
    ``tests/test_oracle.py``).  Used where the literal form is too slow.

Tie rule: the reference ranks with ``np.argsort(scores)[::-1]`` whose order among
equal scores is implementation defined.  The oracle (and the HIP path) fix it as
"score descending, then row index descending", i.e. a stable ascending sort
reversed.
"""
import math

import numpy as np


class BM25OkapiLiteral:
    """rank_bm25 0.2.x ``BM25Okapi`` restated with per-document dicts."""

    def __init__(self, corpus, k1=1.5, b=0.75, epsilon=0.25):
        self.k1, self.b, self.epsilon = k1, b, epsilon
        self.corpus_size = 0
        self.doc_freqs = []
        self.doc_len = []
        self.idf = {}
        df = {}
        total = 0
        for doc in corpus:
            self.doc_len.append(len(doc))
            total += len(doc)
            counts = {}
            for w in doc:
                counts[w] = counts.get(w, 0) + 1
            self.doc_freqs.append(counts)
            for w in counts:
                df[w] = df.get(w, 0) + 1
            self.corpus_size += 1
        self.avgdl = total / self.corpus_size
        # idf with the epsilon floor for negative values
        idf_sum = 0
        negative = []
        for w, n_w in df.items():
            v = math.log(self.corpus_size - n_w + 0.5) - math.log(n_w + 0.5)
            self.idf[w] = v
            idf_sum += v
            if v < 0:
                negative.append(w)
        self.average_idf = idf_sum / len(self.idf)
        floor = self.epsilon * self.average_idf
        for w in negative:
            self.idf[w] = floor

    def get_scores(self, query):
        score = np.zeros(self.corpus_size)
        doc_len = np.array(self.doc_len)
        for q in query:
            q_freq = np.array([(d.get(q) or 0) for d in self.doc_freqs])
            score += (self.idf.get(q) or 0) * (
                q_freq * (self.k1 + 1)
                / (q_freq + self.k1 * (1 - self.b + self.b * doc_len / self.avgdl)))
        return score


class BM25OkapiArrays:
    """Same arithmetic as ``BM25OkapiLiteral`` on term-major posting arrays.

    ``vocab`` maps a term to its id in first-appearance order (the iteration order
    of the upstream document-frequency dict, which fixes the order of the
    ``average_idf`` sum).
    """

    def __init__(self, corpus, k1=1.5, b=0.75, epsilon=0.25):
        self.k1, self.b, self.epsilon = k1, b, epsilon
        self.corpus_size = len(corpus)
        self.vocab = {}
        term_ids = []
        doc_ids = []
        doc_len = np.zeros(self.corpus_size, dtype=np.int64)
        for d, doc in enumerate(corpus):
            doc_len[d] = len(doc)
            for w in doc:
                t = self.vocab.get(w)
                if t is None:
                    t = len(self.vocab)
                    self.vocab[w] = t
                term_ids.append(t)
                doc_ids.append(d)
        self._finish(np.asarray(term_ids, dtype=np.int64), np.asarray(doc_ids, dtype=np.int64), doc_len)

    @classmethod
    def from_token_ids(cls, term_ids, doc_ids, doc_len, vocab_size, k1=1.5, b=0.75, epsilon=0.25):
        """Build from flat (term id, doc id) token occurrences; ids must already be in
        first-appearance order when bit-equality with the literal form is wanted."""
        self = cls.__new__(cls)
        self.k1, self.b, self.epsilon = k1, b, epsilon
        self.corpus_size = int(len(doc_len))
        self.vocab = None
        self._vocab_size = int(vocab_size)
        self._finish(np.asarray(term_ids, dtype=np.int64), np.asarray(doc_ids, dtype=np.int64),
                     np.asarray(doc_len, dtype=np.int64))
        return self

    def _finish(self, term_ids, doc_ids, doc_len):
        n = self.corpus_size
        v = len(self.vocab) if self.vocab is not None else self._vocab_size
        self.doc_len = doc_len
        self.avgdl = int(doc_len.sum()) / n
        # (term, doc) -> tf
        key = term_ids * n + doc_ids
        uniq, tf = np.unique(key, return_counts=True)
        self.post_term = uniq // n
        self.post_doc = uniq % n
        self.post_tf = tf.astype(np.int64)
        self.indptr = np.zeros(v + 1, dtype=np.int64)
        np.add.at(self.indptr, self.post_term + 1, 1)
        self.indptr = np.cumsum(self.indptr)
        df = np.diff(self.indptr)
        # idf in term-id order, summed sequentially like the upstream loop
        self.idf = np.zeros(v)
        idf_sum = 0
        for t in range(v):
            if df[t] == 0:
                continue
            val = math.log(n - int(df[t]) + 0.5) - math.log(int(df[t]) + 0.5)
            self.idf[t] = val
            idf_sum += val
        n_terms = int((df > 0).sum())
        self.average_idf = idf_sum / n_terms
        self.idf[(self.idf < 0) & (df > 0)] = self.epsilon * self.average_idf
        self.df = df

    def term_id(self, token):
        if self.vocab is None:
            raise ValueError("built from ids; pass ids to get_scores_ids")
        return self.vocab.get(token, -1)

    def get_scores(self, query_tokens):
        return self.get_scores_ids([self.term_id(q) for q in query_tokens])

    def get_scores_ids(self, query_ids):
        n = self.corpus_size
        score = np.zeros(n)
        doc_len = self.doc_len
        for t in query_ids:
            q_freq = np.zeros(n, dtype=np.int64)
            idf = 0
            if t is not None and t >= 0:
                lo, hi = self.indptr[t], self.indptr[t + 1]
                q_freq[self.post_doc[lo:hi]] = self.post_tf[lo:hi]
                idf = float(self.idf[t]) or 0
            score += idf * (
                q_freq * (self.k1 + 1)
                / (q_freq + self.k1 * (1 - self.b + self.b * doc_len / self.avgdl)))
        return score


def rank_rows_ref(scores, limit):
    """Row order the oracle assigns to ``np.argsort(scores)[::-1][:limit]``:
    score descending, ties by row index descending."""
    order = np.argsort(scores, kind='stable')[::-1]
    return order[:limit]


def bm25_search_ref(scores, chunks, top_k=10, filters=None):
    """Restates the candidate window + post-filter + formatting of
    ``BM25Indexer.search`` (reference ``src/indexing/bm25_indexer.py:128-163``)."""
    out = []
    for idx in rank_rows_ref(scores, top_k * 2):
        idx = int(idx)
        c = chunks[idx]
        d = c.to_dict() if hasattr(c, 'to_dict') else c
        if filters and any(d.get(k) != v for k, v in filters.items()):
            continue
        out.append({
            'chunk_id': d.get('chunk_id', f'chunk_{idx}'),
            'doc_id': d.get('doc_id', ''),
            'source': d.get('source', ''),
            'chunk_type': d.get('chunk_type', ''),
            'text': d.get('text', ''),
            'metadata': d.get('metadata', {}),
            'score': float(scores[idx]),
        })
        if len(out) >= top_k:
            break
    return out
