"""Seeded synthetic inputs shared by the tests (small) and bench.py (full size)."""
import numpy as np


def dense_corpus(n, dim, seed=0, dup_frac=0.0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, dim), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    if dup_frac > 0 and n > 1:
        m = max(1, int(n * dup_frac))
        dst = rng.choice(np.arange(1, n), size=m, replace=False)
        src = (rng.random(m) * dst).astype(np.int64)       # an earlier row
        x[dst] = x[src]
    return x.astype(np.float32)


def dense_queries(corpus, b, seed=1, noise=0.5, random_frac=0.1):
    rng = np.random.default_rng(seed)
    n, dim = corpus.shape
    q = corpus[rng.integers(0, n, size=b)] + noise * rng.standard_normal((b, dim), dtype=np.float32) / np.sqrt(dim)
    r = rng.random(b) < random_frac
    q[r] = rng.standard_normal((int(r.sum()), dim), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(np.float32)


def zipf_probs(vocab, s=1.07):
    p = 1.0 / np.power(np.arange(1, vocab + 1, dtype=np.float64), s)
    return p / p.sum()


def sparse_corpus(n_docs, vocab, seed=0, mean_len=45.0, sigma=0.6, max_len=128):
    """-> (term_ids, doc_ids, doc_len) flat token occurrences, Zipf terms, log-normal lengths."""
    rng = np.random.default_rng(seed)
    ln = np.clip(np.rint(rng.lognormal(np.log(mean_len), sigma, n_docs)), 2, max_len).astype(np.int64)
    total = int(ln.sum())
    cdf = np.cumsum(zipf_probs(vocab))
    term = np.searchsorted(cdf, rng.random(total), side='left').astype(np.int64)
    np.minimum(term, vocab - 1, out=term)
    doc = np.repeat(np.arange(n_docs, dtype=np.int64), ln)
    return term, doc, ln


def sparse_queries(term, doc, doc_len, vocab, b, seed=1, oov_frac=0.01):
    """-> list of term-id lists; 70 % of the tokens from a random document, 30 % from the prior."""
    rng = np.random.default_rng(seed)
    cdf = np.cumsum(zipf_probs(vocab))
    start = np.concatenate([[0], np.cumsum(doc_len)])
    out = []
    for _ in range(b):
        n = int(rng.integers(3, 13))
        if rng.random() < oov_frac:
            out.append([-1] * n)
            continue
        d = int(rng.integers(0, len(doc_len)))
        q = []
        for _ in range(n):
            if rng.random() < 0.7:
                q.append(int(term[start[d] + rng.integers(0, doc_len[d])]))
            else:
                q.append(int(min(np.searchsorted(cdf, rng.random()), vocab - 1)))
        out.append(q)
    return out


def remap_first_appearance(term, vocab):
    """Renumber term ids in order of first appearance (the id order a dict-based builder
    assigns); returns (new ids, old->new map with -1 for absent terms, number of terms)."""
    uniq, first = np.unique(term, return_index=True)
    order = uniq[np.argsort(first, kind='stable')]
    mapping = np.full(vocab, -1, np.int64)
    mapping[order] = np.arange(order.shape[0])
    return mapping[term], mapping, int(order.shape[0])
