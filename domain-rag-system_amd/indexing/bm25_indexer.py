"""BM25 retriever backed by the HIP postings kernels.

Drop-in for the reference ``BM25Indexer`` (reference ``src/indexing/bm25_indexer.py:20-213``):
same constructor, ``tokenize`` / ``index_chunks`` / ``search`` / ``save`` / ``load`` /
``get_stats`` and the same result dicts.  What differs is underneath: instead of a
``rank_bm25.BM25Okapi`` object holding one dict per document, the index is a term-major CSR
(``indptr``, ``rows``, fp64 ``impact``) plus ``idf``, resident in HBM, and ``search`` runs
``rag_bm25_search`` (no CPU scoring path exists; a missing library or GPU raises).

Arithmetic contract (rank_bm25 0.2.x ``BM25Okapi``, restated):
    idf(t)   = ln(N - df + 0.5) - ln(df + 0.5); negative values become 0.25 * mean idf
    impact   = tf*(k1+1) / (tf + k1*(1 - b + b*doc_len/avgdl))
    score(d) = sum over query tokens, in query order, of idf(t) * impact(t, d)     (fp64)
Ranking: score descending, equal scores by row index descending (the order a stable
``argsort`` followed by ``[::-1]`` produces; upstream's order among ties is unspecified).
"""
import io
import math
import pickle
import re
import warnings
from pathlib import Path

import numpy as np

try:                                    # both import spellings of the reference tree work
    from preprocessing import MedicalTermNormalizer
except ImportError:                     # pragma: no cover - depends on sys.path layout
    from ..preprocessing import MedicalTermNormalizer

import rag_hip

_TOKEN = re.compile(r'\b\w+\b')
_EPSILON = 0.25


class _ForeignObject:
    """Inert stand-in for an instance of a class this package does not have (a pickled
    ``rank_bm25.BM25Okapi``, a chunk object of another package): it keeps the instance's attribute
    dictionary and runs none of the original class's code."""

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2:          # (dict, slots)
            for part in state:
                if isinstance(part, dict):
                    self.__dict__.update(part)

    def to_dict(self):                                               # chunk objects
        return {k: self.__dict__.get(k, d) for k, d in (('chunk_id', ''), ('doc_id', ''), ('source', ''),
                                                        ('chunk_type', ''), ('text', ''), ('metadata', {}))}


class _IndexUnpickler(pickle.Unpickler):
    """Unpickler for ``bm25_index.pkl`` that executes nothing from the file: only builtin containers,
    numpy array / scalar reconstruction and (as inert attribute bags) the foreign classes a pickle written by
    the reference holds (``rank_bm25.BM25Okapi``, ``...chunker.Chunk``).  Any other global is refused."""

    _SAFE = {
        ('builtins', n) for n in ('dict', 'list', 'tuple', 'set', 'frozenset', 'int', 'float', 'complex', 'str',
                                  'bytes', 'bytearray', 'bool', 'slice', 'range', 'object')
    } | {('collections', 'OrderedDict'), ('collections', 'defaultdict'), ('copyreg', '_reconstructor')}
    _NUMPY = {'_reconstruct', 'scalar', 'ndarray', 'dtype', '_frombuffer'}

    def find_class(self, module, name):
        if (module, name) in self._SAFE:
            return super().find_class(module, name)
        if module in ('numpy', 'numpy.core.multiarray', 'numpy._core.multiarray', 'numpy.core.numeric',
                      'numpy._core.numeric') and name in self._NUMPY:
            return super().find_class(module, name)
        if module.startswith('numpy') and name.startswith(('float', 'int', 'uint', 'bool')):     # numpy.float64 ...
            return super().find_class(module, name)
        if (module.split('.')[0] == 'rank_bm25' and name.startswith('BM25')) or (module.endswith('chunker') and name == 'Chunk'):
            return _ForeignObject
        raise pickle.UnpicklingError(f"bm25 index file refers to {module}.{name}, which this loader does not admit "
                                     "(only plain containers, numpy arrays and the reference's BM25Okapi / Chunk objects)")


class Bm25Arrays:
    """Host-side image of the device index (what save/load persist)."""

    def __init__(self, vocab, indptr, rows, tf, idf, doc_len, avgdl, k1, b):
        self.vocab, self.indptr, self.rows, self.tf = vocab, indptr, rows, tf
        self.idf, self.doc_len, self.avgdl, self.k1, self.b = idf, doc_len, avgdl, k1, b

    @property
    def n_docs(self):
        return int(self.doc_len.shape[0])

    def impacts(self):
        """tf*(k1+1) / (tf + k1*(1 - b + b*doc_len/avgdl)), evaluated in the operand order of
        the upstream expression so every posting's value is bit-identical to it."""
        if getattr(self, '_impacts', None) is not None:      # computed on the device at build time
            return self._impacts
        k1, b = self.k1, self.b
        norm = k1 * (1 - b + b * self.doc_len / self.avgdl)
        tf = self.tf.astype(np.int64)
        return tf * (k1 + 1) / (tf + norm[self.rows])


def build_arrays(term_ids, doc_ids, doc_len, vocab, k1, b, global_stats=None):
    """Flat token occurrences (term id, doc id) -> CSR + statistics.

    ``global_stats`` (row-sharded corpora): ``{'df': int64[V], 'total_len': int, 'n_docs': int}``
    summed over all shards; idf and avgdl then come from the whole corpus, so every document
    scores exactly as it would in an unsharded index, while the postings stay local."""
    n_docs = int(doc_len.shape[0])
    v = len(vocab) if not isinstance(vocab, int) else vocab
    if n_docs == 0:
        raise ValueError("cannot build a BM25 index over zero documents")
    key = term_ids.astype(np.int64) * n_docs + doc_ids.astype(np.int64)
    uniq, tf = np.unique(key, return_counts=True)
    post_term = uniq // n_docs
    rows = (uniq - post_term * n_docs).astype(np.uint32)
    df = np.bincount(post_term, minlength=v).astype(np.int64)
    indptr = np.zeros(v + 1, np.int64)
    np.cumsum(df, out=indptr[1:])
    n_local = n_docs
    total_len = int(doc_len.sum())
    if global_stats is not None:
        df = np.asarray(global_stats['df'], np.int64)
        n_docs = int(global_stats['n_docs'])
        total_len = int(global_stats['total_len'])
    idf = _idf(df, n_docs, v)
    avgdl = total_len / n_docs
    assert doc_len.shape[0] == n_local
    return Bm25Arrays(vocab, indptr, rows, tf.astype(np.uint32), idf, doc_len.astype(np.int64), avgdl, k1, b)


def _idf(df, n_docs, v):
    """idf with libm's log, summed in vocabulary (first appearance) order; negative values are
    replaced by epsilon * mean idf, as rank_bm25's BM25Okapi does."""
    idf = np.zeros(v, np.float64)
    total = 0
    log = math.log
    for t in range(v):
        n_t = int(df[t])
        if n_t:
            val = log(n_docs - n_t + 0.5) - log(n_t + 0.5)
            idf[t] = val
            total += val
    present = int(np.count_nonzero(df))
    if present:
        floor = _EPSILON * (total / present)
        idf[(idf < 0) & (df > 0)] = floor
    return idf


def build_arrays_device(term_ids, doc_ids, doc_len, vocab, k1, b, global_stats=None, device='cuda'):
    """build_arrays with the counting, the sort into term-major order and the impacts done on the
    GPU (rag_bm25_build_dev, rag_bm25_impacts_dev); same arguments, same result bit for bit.  The
    occurrences must be grouped by document in ascending order (they are when documents are
    tokenised one after the other); otherwise they are stably sorted by document first."""
    import torch
    n_docs = int(doc_len.shape[0])
    v = len(vocab) if not isinstance(vocab, int) else vocab
    if n_docs == 0:
        raise ValueError("cannot build a BM25 index over zero documents")
    term_ids = np.asarray(term_ids)
    doc_ids = np.asarray(doc_ids)
    if doc_ids.size and np.any(np.diff(doc_ids) < 0):
        order = np.argsort(doc_ids, kind='stable')
        term_ids, doc_ids = term_ids[order], doc_ids[order]
    doc_off = np.zeros(n_docs + 1, np.int64)
    np.cumsum(np.bincount(doc_ids, minlength=n_docs), out=doc_off[1:])
    dev = torch.device(device)
    term_d = torch.from_numpy(np.ascontiguousarray(term_ids, np.int32)).to(dev)
    indptr_d, rows_d, tf_d = rag_hip.bm25_build_dev(term_d, torch.from_numpy(doc_off).to(dev), v)
    indptr = indptr_d.cpu().numpy()
    df = np.diff(indptr).astype(np.int64)
    total_len = int(doc_len.sum())
    n_global = n_docs
    if global_stats is not None:
        df = np.asarray(global_stats['df'], np.int64)
        n_global = int(global_stats['n_docs'])
        total_len = int(global_stats['total_len'])
    idf = _idf(df, n_global, v)
    avgdl = total_len / n_global
    doc_len64 = np.ascontiguousarray(doc_len, np.int64)
    impacts = rag_hip.bm25_impacts_dev(rows_d, tf_d, torch.from_numpy(doc_len64).to(dev), avgdl, k1, b)
    arr = Bm25Arrays(vocab, indptr, rows_d.cpu().numpy().view(np.uint32), tf_d.cpu().numpy().view(np.uint32), idf,
                     doc_len64, avgdl, k1, b)
    arr._impacts = impacts.cpu().numpy()
    return arr


class BM25Indexer:
    def __init__(self, k1: float = 1.5, b: float = 0.75, medical_normalizer: MedicalTermNormalizer = None):
        rag_hip.load_library()              # fail at construction when the backend is missing
        self.k1 = k1
        self.b = b
        self.normalizer = medical_normalizer or MedicalTermNormalizer()
        self.bm25 = None                    # Bm25Arrays once built (name kept for callers)
        self.chunks = []
        self._tokenized_corpus = []
        self._doc_tokens = None
        self._words = []
        self._device_index = None
        self._short_keys = None

    # ---- tokenizer -------------------------------------------------------------------
    def tokenize(self, text: str):
        """Lower-case word tokens of at least two characters; a one-character token survives
        only when it equals a lower-cased abbreviation key occurring in the text."""
        low = text.lower()
        tokens = _TOKEN.findall(low)
        if self._short_keys is None:
            self._short_keys = [k.lower() for k in self.normalizer.abbreviations.keys() if len(k.lower()) < 2]
        if not self._short_keys:            # no abbreviation key can rescue a short token
            return [t for t in tokens if len(t) >= 2]
        rescued = {k for k in self._short_keys if k in low}
        return [t for t in tokens if len(t) >= 2 or t in rescued]

    # ---- build -----------------------------------------------------------------------
    @staticmethod
    def _text_of(chunk):
        return chunk.text if hasattr(chunk, 'text') else chunk['text']

    def index_chunks(self, chunks):
        """Tokenise every chunk and build the device index.  Rebuilds from scratch on every
        call (the reference appends to its token lists and corrupts a second build).  Tokenizer and
        vocabulary run natively (rag_bm25_tokenize_*): ASCII texts are tokenised there, any other text by
        `tokenize` here (Unicode word classes and case folding are Python's) and handed over as tokens."""
        self.chunks = chunks
        docs, flags = [], np.zeros(len(chunks), np.uint8)
        for d, c in enumerate(chunks):
            text = self._text_of(c)
            if text.isascii():
                docs.append(text.encode('ascii'))
            else:
                docs.append('\n'.join(self.tokenize(text)).encode('utf-8'))
                flags[d] = 1
        self._build_native(docs, flags)

    def _short_key_string(self):
        if self._short_keys is None:
            self._short_keys = [k.lower() for k in self.normalizer.abbreviations.keys() if len(k.lower()) < 2]
        return ''.join(k for k in self._short_keys if len(k) == 1 and k.isascii())

    def _build(self, tokenized):
        """Build from token lists (a loaded index file's `tokenized_corpus`)."""
        docs = ['\n'.join(doc).encode('utf-8') for doc in tokenized]
        self._build_native(docs, np.ones(len(docs), np.uint8))

    def _build_native(self, docs, flags):
        term_ids, doc_len, words = rag_hip.bm25_tokenize(docs, flags, self._short_key_string())
        vocab = {w: i for i, w in enumerate(words)}
        self._words = words
        self._doc_tokens = (term_ids, np.concatenate([[0], np.cumsum(doc_len)]))
        self._tokenized_corpus = None
        doc_ids = np.repeat(np.arange(len(docs), dtype=np.int64), doc_len)
        # the counting and the sort into term-major order run on the GPU (same arrays as build_arrays)
        self.bm25 = build_arrays_device(term_ids, doc_ids, doc_len, vocab, self.k1, self.b)
        self._upload()

    @property
    def tokenized_corpus(self):
        """Token lists of the indexed chunks (the reference keeps them and pickles them); materialised from the
        native build's id arrays on first use."""
        if self._tokenized_corpus is None:
            if self._doc_tokens is None:
                return []
            ids, off = self._doc_tokens
            words = self._words
            self._tokenized_corpus = [[words[t] for t in ids[off[d]:off[d + 1]]] for d in range(len(off) - 1)]
        return self._tokenized_corpus

    @tokenized_corpus.setter
    def tokenized_corpus(self, value):
        self._tokenized_corpus = value
        self._doc_tokens = None

    def _upload(self):
        a = self.bm25
        if self._device_index is not None:
            self._device_index.close()
        self._device_index = rag_hip.Bm25Index(a.indptr, a.rows, a.impacts(), a.idf, a.n_docs)

    # ---- search ----------------------------------------------------------------------
    def term_ids(self, tokens):
        vocab = self.bm25.vocab
        return [vocab.get(t, -1) for t in tokens]

    def search_rows(self, queries, k):
        """Batched core: list of query strings -> (scores [B,k] f64, rows [B,k], counts)."""
        if self.bm25 is None:
            raise ValueError("Index not built. Call index_chunks() first.")
        ids = [self.term_ids(self.tokenize(q)) for q in queries]
        if int(k) > rag_hip.MAX_K:
            warnings.warn(f"BM25 search asked for {int(k)} results per query; the kernels return at most "
                          f"{rag_hip.MAX_K}, so the list is cut there", RuntimeWarning, stacklevel=2)
        k = min(int(k), rag_hip.MAX_K)
        out = self._device_index.search(ids, k)
        rag_hip.check_faults()               # a kernel that met inconsistent data flags it instead of faulting
        return out

    def _format(self, idx, score):
        c = self.chunks[idx]
        d = c.to_dict() if hasattr(c, 'to_dict') else c
        return {
            'chunk_id': d.get('chunk_id', f'chunk_{idx}'),
            'doc_id': d.get('doc_id', ''),
            'source': d.get('source', ''),
            'chunk_type': d.get('chunk_type', ''),
            'text': d.get('text', ''),
            'metadata': d.get('metadata', {}),
            'score': score,
        }

    def search(self, query: str, top_k: int = 10, filters: dict = None):
        """Best ``top_k`` chunks out of a ``2*top_k`` candidate window; ``filters`` are
        equality tests on top-level chunk keys applied inside that window, so fewer than
        ``top_k`` results can come back (reference ``bm25_indexer.py:128-161``)."""
        if self.bm25 is None:
            raise ValueError("Index not built. Call index_chunks() first.")
        if top_k <= 0:
            return []
        window = top_k * 2 if filters else top_k
        score, row, cnt = self.search_rows([query], min(window, max(len(self.chunks), 1)))
        out = []
        for j in range(int(cnt[0])):
            idx = int(row[0, j])
            r = self._format(idx, float(score[0, j]))
            if filters:
                c = self.chunks[idx]
                d = c.to_dict() if hasattr(c, 'to_dict') else c
                if any(d.get(key) != val for key, val in filters.items()):
                    continue
            out.append(r)
            if len(out) >= top_k:
                break
        return out

    # ---- persistence -----------------------------------------------------------------
    def save(self, filepath: str):
        """``bm25_index.pkl``: a pickle of a plain dict with the reference's five keys
        (``bm25``, ``chunks``, ``tokenized_corpus``, ``k1``, ``b``).  ``bm25`` holds plain
        numpy arrays / builtins instead of a ``rank_bm25`` instance."""
        filepath = Path(filepath)
        filepath.parent.mkdir(parents=True, exist_ok=True)
        a = self.bm25
        state = None if a is None else {
            'format': 'rag_hip.bm25.v1', 'vocab': a.vocab, 'indptr': a.indptr, 'rows': a.rows, 'tf': a.tf,
            'idf': a.idf, 'doc_len': a.doc_len, 'avgdl': a.avgdl, 'k1': a.k1, 'b': a.b, 'epsilon': _EPSILON,
            'corpus_size': a.n_docs}
        chunks = [c.to_dict() if hasattr(c, 'to_dict') else c for c in self.chunks]
        with open(filepath, 'wb') as f:
            pickle.dump({'bm25': state, 'chunks': chunks, 'tokenized_corpus': self.tokenized_corpus,
                         'k1': self.k1, 'b': self.b}, f, protocol=pickle.HIGHEST_PROTOCOL)

    def load(self, filepath: str):
        filepath = Path(filepath)
        if not filepath.exists():
            raise FileNotFoundError(f"Index file not found: {filepath}")
        with open(filepath, 'rb') as f:
            # executes nothing from the file: see _IndexUnpickler
            data = _IndexUnpickler(io.BytesIO(f.read())).load()
        if not isinstance(data, dict) or not {'chunks', 'tokenized_corpus', 'k1', 'b'} <= set(data):
            raise ValueError(f"{filepath} is not a BM25 index file (keys 'bm25', 'chunks', 'tokenized_corpus', 'k1', 'b')")
        self.chunks = [c.to_dict() if isinstance(c, _ForeignObject) else c for c in data['chunks']]
        self.tokenized_corpus = data['tokenized_corpus']
        self.k1 = data['k1']
        self.b = data['b']
        s = data['bm25']
        if isinstance(s, dict) and s.get('format') == 'rag_hip.bm25.v1':
            self.bm25 = Bm25Arrays(s['vocab'], s['indptr'], s['rows'], s['tf'], s['idf'], s['doc_len'],
                                   s['avgdl'], s['k1'], s['b'])
            self._upload()
        elif self.tokenized_corpus:
            # a file the reference wrote (its 'bm25' entry is a pickled rank_bm25.BM25Okapi, loaded as an inert
            # attribute bag): the index is rebuilt from the token lists the file carries, which is what that
            # object was built from (reference src/indexing/bm25_indexer.py:102-106, :170-176)
            self._build(self.tokenized_corpus)
        else:
            self.bm25 = None

    def get_stats(self):
        if self.bm25 is None:
            return {'error': 'Index not built'}
        a = self.bm25
        return {
            'num_documents': len(self.chunks),
            'avg_doc_length': float(np.mean(a.doc_len)) if a.n_docs else 0.0,
            'vocab_size': int(np.count_nonzero(np.diff(a.indptr))),
            'k1': self.k1,
            'b': self.b,
        }
