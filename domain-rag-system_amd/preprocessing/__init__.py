"""Query-side text helpers the retrieval path needs (no scispaCy, no corpus loaders).

Only the pieces the hot path touches are provided: the abbreviation table read by the BM25
tokenizer, the ``Chunk`` record, and a rule-based query preprocessor.  Document loading,
chunking and NER of the reference's ``src/preprocessing`` are offline data preparation and
are not part of this package.
"""
from .medical_term_normalizer import MedicalTermNormalizer
from .query_preprocessor import QueryPreprocessor
from .chunk import Chunk

__all__ = ['MedicalTermNormalizer', 'QueryPreprocessor', 'Chunk']
