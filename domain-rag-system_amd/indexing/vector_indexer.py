"""Dense retriever backed by the HIP exact-kNN kernels.

Drop-in for the reference ``VectorIndexer`` (reference ``src/indexing/vector_indexer.py:21-249``):
same constructor, ``create_collection`` / ``index_chunks`` / ``search`` / ``get_stats`` /
``delete_collection`` and the same result dicts.  The embedded Qdrant client is replaced by
an ``[N x dim]`` float32 matrix resident in HBM (``rag_hip.DenseIndex``) plus the payload
dicts on the host.  Semantics restated from qdrant-client's local mode:

* ``cosine``: rows are L2-normalised (float32) when stored and the query when searched;
  score = float32 dot product.  ``dot``: raw dot product.  ``euclidean``: distance,
  smaller is better.
* exact search over all points passing the payload filter (every ``key == value`` must
  hold; dotted keys reach into nested dicts), best first, ``limit`` results,
  ``score_threshold`` keeps scores at least as good.
* equal scores: row index descending.

On-disk layout under ``storage_path/<collection_name>/`` (the reference's directory is a
Qdrant sqlite store, which cannot be read or written without qdrant-client):
``embeddings.f32`` (row-major), ``payload.jsonl`` (one payload dict per row),
``canon.u32`` (first row carrying the same chunk_id, used by rank fusion) and
``meta.json``.
"""
import json
import shutil
from pathlib import Path
from typing import Dict, List

import numpy as np

import rag_hip
from .embedder import MedicalEmbedder

_PAYLOAD_KEYS = ('chunk_id', 'doc_id', 'source', 'chunk_type', 'text', 'metadata')


def _lookup(payload, dotted):
    cur = payload
    for part in dotted.split('.'):
        if not isinstance(cur, dict) or part not in cur:
            return _MISSING
        cur = cur[part]
    return cur


_MISSING = object()


def canonical_rows(chunk_ids):
    """canon[row] = first row with the same chunk_id (chunk ids need not be unique)."""
    first = {}
    canon = np.empty(len(chunk_ids), np.uint32)
    for i, cid in enumerate(chunk_ids):
        canon[i] = first.setdefault(cid, i)
    return canon


class VectorIndexer:
    def __init__(self, collection_name: str = 'medical_docs', embedder: MedicalEmbedder = None,
                 storage_path: str = './data/vector_db', distance_metric: str = 'cosine',
                 use_in_memory: bool = False):
        rag_hip.load_library()              # fail at construction when the backend is missing
        self.collection_name = collection_name
        self.embedder = embedder or MedicalEmbedder()
        self.use_in_memory = use_in_memory
        self.storage_path = None if use_in_memory else Path(storage_path)
        if self.storage_path is not None:
            self.storage_path.mkdir(parents=True, exist_ok=True)
        metric = distance_metric.lower()
        self.distance_metric = metric if metric in ('cosine', 'euclidean', 'dot') else 'cosine'
        self._index = None                  # rag_hip.DenseIndex
        self._payloads = []
        self._canon = None
        self._code_cols, self._row_filter = {}, None   # per-key categorical codes for device-side filters
        self._text_rows = None              # text -> first row holding it (rows_of_texts)
        self.rows_are_embeddings = False    # the rows are this embedder's embeddings of the payload texts (index_chunks)
        self._created = False

    # ---- store management ------------------------------------------------------------
    def _dir(self):
        return None if self.storage_path is None else self.storage_path / self.collection_name

    def _drop(self):
        if self._index is not None:
            self._index.close()
        self._index, self._payloads, self._canon = None, [], None
        self._code_cols, self._row_filter = {}, None
        self._text_rows = None

    def create_collection(self, force_recreate: bool = True):
        d = self._dir()
        exists = self._index is not None or (d is not None and (d / 'meta.json').exists())
        if exists and force_recreate:
            self._drop()
            if d is not None and d.exists():
                shutil.rmtree(d)
            exists = False
        if not exists and d is not None:
            d.mkdir(parents=True, exist_ok=True)
        self._created = True

    def delete_collection(self):
        self._drop()
        d = self._dir()
        if d is not None and d.exists():
            shutil.rmtree(d)
        self._created = False

    def _attach(self):
        """Load a persisted collection on first use."""
        if self._index is not None:
            return True
        d = self._dir()
        if d is None or not (d / 'meta.json').exists():
            return False
        with open(d / 'meta.json', 'r', encoding='utf-8') as f:
            meta = json.load(f)
        n, dim = meta['points_count'], meta['embedding_dim']
        rows = np.fromfile(d / 'embeddings.f32', dtype=np.float32).reshape(n, dim)
        with open(d / 'payload.jsonl', 'r', encoding='utf-8') as f:
            self._payloads = [json.loads(line) for line in f]
        self._code_cols, self._row_filter = {}, None
        self._canon = np.fromfile(d / 'canon.u32', dtype=np.uint32)
        self.distance_metric = meta.get('distance_metric', self.distance_metric)
        self._index = rag_hip.DenseIndex(rows, metric=self.distance_metric)
        # rows count as "this embedder's embedding of the payload text" only if index_chunks wrote them with this model
        self.rows_are_embeddings = bool(meta.get('rows_are_embeddings')) and \
            meta.get('model_name') == getattr(self.embedder, 'model_name', None)
        return True

    # ---- build -----------------------------------------------------------------------
    def index_chunks(self, chunks: List, batch_size: int = 100):
        """Embed every chunk text and store one point per chunk (id = row index).
        ``batch_size`` is accepted for signature compatibility; the matrix is uploaded once."""
        texts = [c.text if hasattr(c, 'text') else c['text'] for c in chunks]
        emb = self.embedder.encode(texts) if texts else np.zeros((0, self.embedder.embedding_dim), np.float32)
        payloads = []
        for idx, c in enumerate(chunks):
            d = c.to_dict() if hasattr(c, 'to_dict') else c
            payloads.append({
                'chunk_id': d.get('chunk_id', f'chunk_{idx}'),
                'doc_id': d.get('doc_id', ''),
                'source': d.get('source', ''),
                'chunk_type': d.get('chunk_type', ''),
                'text': d.get('text', ''),
                'metadata': d.get('metadata', {}),
            })
        self.add_points(emb, payloads, _embedded_here=True)

    def add_points(self, vectors: np.ndarray, payloads: List[Dict], _embedded_here: bool = False):
        """Store precomputed vectors (replaces the collection's content).  Vectors handed in from outside are not
        known to be this embedder's embeddings of the payload texts: only index_chunks marks the rows as such
        (`rows_are_embeddings`, which the simple reranker's resident-row route requires)."""
        self.rows_are_embeddings = bool(_embedded_here)
        rows = np.ascontiguousarray(vectors, dtype=np.float32)
        if rows.ndim != 2 or rows.shape[0] != len(payloads):
            raise ValueError("need one vector per payload")
        if self.distance_metric == 'cosine' and rows.shape[0]:
            norm = np.linalg.norm(rows, axis=1, keepdims=True).astype(np.float32)
            rows = (rows / np.where(norm == 0, np.float32(1), norm)).astype(np.float32)
        self._drop()
        self._payloads = list(payloads)
        self._code_cols, self._row_filter = {}, None
        self._canon = canonical_rows([p['chunk_id'] for p in payloads])
        self._index = rag_hip.DenseIndex(rows, metric=self.distance_metric)
        d = self._dir()
        if d is not None:
            d.mkdir(parents=True, exist_ok=True)
            rows.tofile(d / 'embeddings.f32')
            self._canon.tofile(d / 'canon.u32')
            with open(d / 'payload.jsonl', 'w', encoding='utf-8') as f:
                for p in payloads:
                    f.write(json.dumps(p, ensure_ascii=False) + '\n')
            with open(d / 'meta.json', 'w', encoding='utf-8') as f:
                json.dump({'format': 'rag_hip.dense.v1', 'points_count': int(rows.shape[0]),
                           'embedding_dim': int(rows.shape[1]), 'distance_metric': self.distance_metric,
                           'model_name': self.embedder.model_name,
                           'rows_are_embeddings': self.rows_are_embeddings}, f, indent=2)

    # ---- search ----------------------------------------------------------------------
    def row_mask(self, filters: Dict):
        """Host evaluation of the payload filter (every ``key == value`` must hold): one byte per row."""
        if not filters:
            return None
        mask = np.ones(len(self._payloads), np.uint8)
        for key, value in filters.items():
            for i, p in enumerate(self._payloads):
                if mask[i] and _lookup(p, key) != value:
                    mask[i] = 0
        return mask

    def _code_column(self, key):
        """(value -> code, int32 code per row) of one payload key, built on first use; None when
        the key's values cannot be dictionary keys (the filter then runs on the host)."""
        if key not in self._code_cols:
            table, col = {}, np.empty(len(self._payloads), np.int32)
            try:
                for i, p in enumerate(self._payloads):
                    v = _lookup(p, key)
                    if v is _MISSING:
                        col[i] = rag_hip.FILTER_MISSING
                    elif v != v:                       # NaN equals nothing, itself included
                        col[i] = rag_hip.FILTER_MISSING
                    else:
                        col[i] = table.setdefault(v, len(table))
                self._code_cols[key] = (table, col)
            except TypeError:
                self._code_cols[key] = None
            self._row_filter = None
        return self._code_cols[key]

    def row_mask_dev(self, filters: Dict):
        """The same filter evaluated on the device from per-row categorical codes
        (rag_filter_mask_dev); a uint8 tensor on the index's device, or None without filters."""
        import torch
        if not filters:
            return None
        want = {}
        for key, value in filters.items():
            col = self._code_column(key)
            try:
                want[key] = rag_hip.FILTER_NONE if col is None or value != value else col[0].get(value, rag_hip.FILTER_NONE)
            except TypeError:
                col = None
            if col is None:                            # unhashable values: evaluate on the host
                dev = 'cuda' if torch.cuda.is_available() else 'cpu'
                return torch.from_numpy(self.row_mask(filters)).to(dev)
        if self._row_filter is None or any(k not in self._row_filter.keys for k in want):
            keys = [k for k, v in self._code_cols.items() if v is not None]
            keys = [k for k in keys if k in want] + [k for k in keys if k not in want]
            keys = keys[:rag_hip.RowFilter.MAX_FIELDS]
            if any(k not in keys for k in want):
                dev = 'cuda' if torch.cuda.is_available() else 'cpu'
                return torch.from_numpy(self.row_mask(filters)).to(dev)
            dev = 'cuda' if torch.cuda.is_available() else 'cpu'
            self._row_filter = rag_hip.RowFilter({k: self._code_cols[k][1] for k in keys}, device=dev)
        return self._row_filter.mask_dev(want)

    def prepare_queries(self, vectors: np.ndarray) -> np.ndarray:
        q = np.ascontiguousarray(vectors, dtype=np.float32)
        if self.distance_metric == 'cosine':
            norm = np.linalg.norm(q, axis=1, keepdims=True).astype(np.float32)
            q = (q / np.where(norm == 0, np.float32(1), norm)).astype(np.float32)
        return q

    def search_rows(self, query_vectors: np.ndarray, k: int, filters: Dict = None):
        """Batched core: [B, dim] query vectors -> (scores [B,k] f32, rows [B,k], counts)."""
        if not self._attach():
            raise ValueError(f"Collection {self.collection_name} not found")
        if int(k) > rag_hip.MAX_K:
            import warnings
            warnings.warn(f"vector search asked for {int(k)} results per query; the kernels return at most "
                          f"{rag_hip.MAX_K}, so the list is cut there", RuntimeWarning, stacklevel=2)
        k = min(int(k), rag_hip.MAX_K)
        if not filters:
            out = self._index.search(self.prepare_queries(query_vectors), k)
        else:
            import torch
            mask = self.row_mask_dev(filters)
            q = torch.from_numpy(self.prepare_queries(query_vectors)).to(mask.device)
            score, row, cnt = self._index.search_dev(q, k, row_mask=mask)
            out = score.cpu().numpy(), row.cpu().numpy(), cnt.cpu().numpy()
        rag_hip.check_faults()               # a kernel that met inconsistent data flags it instead of faulting
        return out

    def rows_of_texts(self, texts: List[str]) -> List[int]:
        """Row that holds each text's embedding, -1 for a text the collection does not hold.  Rows with equal
        text have equal embeddings (the embedder is deterministic), so the first such row stands for all."""
        if not self._attach():
            return [-1] * len(texts)
        if self._text_rows is None:
            table = {}
            for i, p in enumerate(self._payloads):
                table.setdefault(p['text'], i)
            self._text_rows = table
        return [self._text_rows.get(t, -1) for t in texts]

    def cosine_to_rows(self, query_vector: np.ndarray, rows: List[int]) -> np.ndarray:
        """Cosine similarity between one query vector and the stored embeddings of `rows` (float32), computed
        from the resident matrix (rag_dense_rescore).  Only for cosine collections, whose rows are stored
        L2-normalised: the float32 dot product with the normalised query is the cosine."""
        if self.distance_metric != 'cosine':
            raise ValueError("cosine_to_rows needs a cosine collection (rows stored L2-normalised)")
        q = self.prepare_queries(np.asarray(query_vector, np.float32).reshape(1, -1))
        out = self._index.rescore(q, np.asarray(rows, np.int64).reshape(1, -1))[0]
        rag_hip.check_faults()
        return out

    def format_hit(self, row: int, score: float) -> Dict:
        p = self._payloads[row]
        return {'chunk_id': p['chunk_id'], 'doc_id': p['doc_id'], 'source': p['source'],
                'chunk_type': p['chunk_type'], 'text': p['text'], 'metadata': p['metadata'], 'score': score}

    def search(self, query: str, top_k: int = 10, filters: Dict = None, score_threshold: float = None) -> List[Dict]:
        if top_k <= 0:
            return []
        score, row, cnt = self.search_rows(self.embedder.encode(query)[:1], top_k, filters)
        out = []
        for j in range(int(cnt[0])):
            s = float(score[0, j])
            if score_threshold is not None:
                if (s > score_threshold) if self.distance_metric == 'euclidean' else (s < score_threshold):
                    break
            out.append(self.format_hit(int(row[0, j]), s))
        return out

    def get_stats(self) -> Dict:
        try:
            if not self._attach():
                raise ValueError(f"Collection {self.collection_name} not found")
            return {
                'collection_name': self.collection_name,
                'points_count': self._index.n_rows,
                'embedding_dim': self.embedder.embedding_dim,
                'model_name': self.embedder.model_name,
                'storage_path': str(self.storage_path) if self.storage_path else 'in-memory',
            }
        except Exception as e:
            return {'error': str(e)}
