"""Rule-based query preprocessing (the scispaCy-free part of the reference's
``QueryPreprocessor``, ``src/preprocessing/query_preprocessor.py:39-103``).

``preprocess`` returns the same dict keys (``original``, ``cleaned``, ``normalized``,
``entities``, ``medical_terms``); ``entities`` stays empty because no NER model is used.
"""
import re

from .medical_term_normalizer import MedicalTermNormalizer

_SPACES = re.compile(r'\s+')
_DROP = re.compile(r'[^\w\s\-\+/().,?]')


class QueryPreprocessor:
    def __init__(self, medical_normalizer: MedicalTermNormalizer = None, use_scispacy: bool = False):
        self.normalizer = medical_normalizer or MedicalTermNormalizer()
        self.nlp = None

    @staticmethod
    def _clean_query(query: str) -> str:
        if not query:
            return ""
        return _DROP.sub('', _SPACES.sub(' ', query)).strip()

    def preprocess(self, query: str, extract_entities: bool = True, normalize_terms: bool = True,
                   expand_abbreviations: bool = True):
        cleaned = self._clean_query(query)
        out = {'original': query, 'cleaned': cleaned, 'normalized': cleaned, 'entities': [], 'medical_terms': {}}
        if normalize_terms:
            out['normalized'] = self.normalizer.normalize_text(cleaned, expand_abbreviations=expand_abbreviations)
            out['medical_terms'] = self.normalizer.extract_medical_terms(cleaned)
        return out

    def get_query_keywords(self, query: str):
        return query.lower().split()
