"""Rerankers: same exports as the reference's ``src/reranker/__init__.py``."""
from .rerank import BaseReranker, build_reranker, SimpleSimilarityReranker, CrossEncoderReranker

__all__ = ['build_reranker', 'SimpleSimilarityReranker', 'CrossEncoderReranker', 'BaseReranker']
