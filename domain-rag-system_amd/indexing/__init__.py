"""Retrieval package: same exports as the reference's ``src/indexing/__init__.py``."""
from .embedder import MedicalEmbedder
from .vector_indexer import VectorIndexer
from .bm25_indexer import BM25Indexer
from .hybrid_indexer import HybridIndexer, HybridRetriever

__all__ = ['MedicalEmbedder', 'VectorIndexer', 'BM25Indexer', 'HybridIndexer', 'HybridRetriever']
