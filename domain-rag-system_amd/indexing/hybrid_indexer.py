"""Hybrid (dense + BM25) retrieval with native rank fusion.

Drop-in for the reference ``HybridIndexer`` (reference ``src/indexing/hybrid_indexer.py:10-246``):
``index_chunks`` builds both indices, ``search`` asks each retriever for ``2*top_k`` results and
fuses them by reciprocal rank fusion (``k = 60``) or min-max weighted sum, ``save_bm25`` /
``load_bm25`` / ``get_stats`` keep their behaviour.  The two fusion methods accept any lists of
result dicts, like the reference's; the arithmetic runs in the ``rag_fuse_*`` kernels on the
chunk ids (interned to integers on the host), so scores are the reference's fp64 values bit for
bit and ties keep first-appearance order.

``HybridRetriever`` is the same class under the name used by the project brief, with
``build_index`` / ``query`` aliases.
"""
from pathlib import Path
from typing import Dict, List

import rag_hip
from .bm25_indexer import BM25Indexer
from .vector_indexer import VectorIndexer


def _intern(results_a, results_b):
    """chunk_id strings -> dense integer ids shared by both lists."""
    table = {}
    out = []
    for results in (results_a, results_b):
        ids = []
        for r in results:
            ids.append(table.setdefault(r['chunk_id'], len(table)))
        out.append(ids)
    return out


def _check_len(*lists):
    for l in lists:
        if len(l) > rag_hip.MAX_FUSE_LIST:
            raise ValueError(f"fusion lists are limited to {rag_hip.MAX_FUSE_LIST} entries (got {len(l)})")


class HybridIndexer:
    def __init__(self, vector_indexer: VectorIndexer = None, bm25_indexer: BM25Indexer = None,
                 storage_path: str = './data/indices'):
        self.storage_path = Path(storage_path)
        self.storage_path.mkdir(parents=True, exist_ok=True)
        self.vector_indexer = vector_indexer
        self.bm25_indexer = bm25_indexer

    # ---- build -----------------------------------------------------------------------
    def index_chunks(self, chunks: List):
        if self.vector_indexer:
            self.vector_indexer.create_collection(force_recreate=True)
            self.vector_indexer.index_chunks(chunks)
        if self.bm25_indexer:
            self.bm25_indexer.index_chunks(chunks)

    # ---- search ----------------------------------------------------------------------
    def search(self, query: str, top_k: int = 10, vector_weight: float = 0.5, bm25_weight: float = 0.5,
               filters: Dict = None, fusion_method: str = 'rrf') -> List[Dict]:
        vector_results = self.vector_indexer.search(query, top_k=top_k * 2, filters=filters) \
            if self.vector_indexer else []
        bm25_results = self.bm25_indexer.search(query, top_k=top_k * 2, filters=filters) \
            if self.bm25_indexer else []
        if fusion_method == 'rrf':
            return self._reciprocal_rank_fusion(vector_results, bm25_results, top_k=top_k)
        return self._weighted_fusion(vector_results, bm25_results, vector_weight=vector_weight,
                                     bm25_weight=bm25_weight, top_k=top_k)

    def _reciprocal_rank_fusion(self, vector_results: List[Dict], bm25_results: List[Dict],
                                top_k: int = 10, k: int = 60) -> List[Dict]:
        """score(id) = sum of 1/(k + rank) over every occurrence of the id (rank from 1 in each
        list); payload from the id's first occurrence; stable descending order."""
        if top_k <= 0 or not (vector_results or bm25_results):
            return []
        _check_len(vector_results, bm25_results)
        v_ids, b_ids = _intern(vector_results, bm25_results)
        pos, _, score, cnt = rag_hip.fuse_rrf([v_ids], [b_ids], min(top_k, rag_hip.MAX_K), k_const=float(k))
        both = list(vector_results) + list(bm25_results)
        fused = []
        for j in range(int(cnt[0])):
            r = both[int(pos[0, j])].copy()
            r['score'] = float(score[0, j])
            r['fusion_method'] = 'rrf'
            fused.append(r)
        return fused

    def _weighted_fusion(self, vector_results: List[Dict], bm25_results: List[Dict], vector_weight: float = 0.5,
                         bm25_weight: float = 0.5, top_k: int = 10) -> List[Dict]:
        """Per-list min-max normalisation (skipped for an empty or constant list), weighted sum
        per id, stable descending order.  Like the reference this writes ``normalized_score``
        into the entries of every list it normalises, and a vector entry supplies the payload
        whenever the id occurs in the vector list."""
        if top_k <= 0 or not (vector_results or bm25_results):
            return []
        _check_len(vector_results, bm25_results)
        for results in (vector_results, bm25_results):
            if results:
                scores = [r['score'] for r in results]
                lo, hi = min(scores), max(scores)
                if hi != lo:
                    for r in results:
                        r['normalized_score'] = (r['score'] - lo) / (hi - lo)
        v_ids, b_ids = _intern(vector_results, bm25_results)
        pos, _, score, cnt = rag_hip.fuse_weighted(
            [v_ids], [[r['score'] for r in vector_results]], [b_ids], [[r['score'] for r in bm25_results]],
            min(top_k, rag_hip.MAX_K), vector_weight=float(vector_weight), bm25_weight=float(bm25_weight))
        both = list(vector_results) + list(bm25_results)
        fused = []
        for j in range(int(cnt[0])):
            r = both[int(pos[0, j])].copy()
            r['score'] = float(score[0, j])
            r['fusion_method'] = 'weighted'
            fused.append(r)
        return fused

    # ---- persistence / stats ---------------------------------------------------------
    def save_bm25(self, filepath: str = None):
        if self.bm25_indexer:
            self.bm25_indexer.save(filepath if filepath is not None else self.storage_path / 'bm25_index.pkl')

    def load_bm25(self, filepath: str = None):
        if self.bm25_indexer:
            self.bm25_indexer.load(filepath if filepath is not None else self.storage_path / 'bm25_index.pkl')

    def get_stats(self) -> Dict:
        stats = {}
        if self.vector_indexer:
            stats['vector'] = self.vector_indexer.get_stats()
        if self.bm25_indexer:
            stats['bm25'] = self.bm25_indexer.get_stats()
        return stats


class HybridRetriever(HybridIndexer):
    """``HybridIndexer`` under the brief's name: ``build_index`` = ``index_chunks``,
    ``query`` = ``search``."""

    def build_index(self, chunks: List):
        return self.index_chunks(chunks)

    def query(self, query: str, top_k: int = 10, **kwargs) -> List[Dict]:
        return self.search(query, top_k=top_k, **kwargs)
