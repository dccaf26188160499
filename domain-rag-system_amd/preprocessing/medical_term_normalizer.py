"""Abbreviation / synonym tables and the text rewriting built on them.

Mirrors the public surface of the reference's ``MedicalTermNormalizer``
(``src/preprocessing/medical_term_normalizer.py:8-296``): ``abbreviations``,
``drug_mapping``, ``condition_synonyms``, ``medical_stopwords``, ``normalize_text``,
``extract_medical_terms``.  The retrieval hot path only reads ``abbreviations`` (BM25
tokenizer) - the rewriting is used by the query CLI before retrieval.
"""
import json
import re
from pathlib import Path

_ABBREVIATIONS = (
    # cardiovascular
    ('MI', 'myocardial infarction'), ('CHF', 'congestive heart failure'), ('HTN', 'hypertension'),
    ('CAD', 'coronary artery disease'), ('AF', 'atrial fibrillation'), ('CVA', 'cerebrovascular accident'),
    # diabetes
    ('DM', 'diabetes mellitus'), ('T1DM', 'type 1 diabetes mellitus'), ('T2DM', 'type 2 diabetes mellitus'),
    ('HbA1c', 'hemoglobin A1c'),
    # respiratory
    ('COPD', 'chronic obstructive pulmonary disease'), ('URI', 'upper respiratory infection'),
    ('SOB', 'shortness of breath'),
    # gastrointestinal
    ('GERD', 'gastroesophageal reflux disease'), ('IBD', 'inflammatory bowel disease'),
    ('IBS', 'irritable bowel syndrome'),
    # general
    ('Rx', 'prescription'), ('Sx', 'symptoms'), ('Tx', 'treatment'), ('Dx', 'diagnosis'), ('Hx', 'history'),
    ('pt', 'patient'), ('w/', 'with'), ('w/o', 'without'),
    # lab values
    ('WBC', 'white blood cell count'), ('RBC', 'red blood cell count'), ('BP', 'blood pressure'),
    ('HR', 'heart rate'), ('BUN', 'blood urea nitrogen'),
    # medications
    ('NSAID', 'nonsteroidal anti-inflammatory drug'), ('ACE', 'angiotensin-converting enzyme'),
    ('ARB', 'angiotensin receptor blocker'), ('SSRI', 'selective serotonin reuptake inhibitor'),
    ('PPI', 'proton pump inhibitor'),
)

_CONDITIONS = (
    ('diabetes mellitus', ('diabetes', 'DM', 'sugar disease')),
    ('hypertension', ('high blood pressure', 'HTN', 'elevated blood pressure')),
    ('myocardial infarction', ('heart attack', 'MI', 'cardiac arrest')),
    ('cerebrovascular accident', ('stroke', 'CVA', 'brain attack')),
    ('gastroesophageal reflux disease', ('GERD', 'acid reflux', 'heartburn')),
    ('chronic obstructive pulmonary disease', ('COPD', 'emphysema', 'chronic bronchitis')),
    ('influenza', ('flu', 'grippe')),
    ('pneumonia', ('lung infection', 'pulmonary infection')),
    ('depression', ('major depressive disorder', 'MDD', 'clinical depression')),
    ('anxiety', ('anxiety disorder', 'GAD', 'generalized anxiety disorder')),
)

_KEEP_WORDS = ('patient', 'treatment', 'drug', 'medication', 'dose', 'side effect', 'symptom', 'disease',
               'condition', 'diagnosis', 'prescription', 'therapy', 'adverse', 'reaction', 'contraindication',
               'warning')


def _load_json(path):
    if path and Path(path).exists():
        with open(path, 'r', encoding='utf-8') as f:
            return json.load(f)
    return None


def _word(term):
    return re.compile(r'\b' + re.escape(term) + r'\b', re.IGNORECASE)


class MedicalTermNormalizer:
    def __init__(self, abbreviations_path: str = None, drug_mapping_path: str = None):
        self.abbreviations = _load_json(abbreviations_path) or dict(_ABBREVIATIONS)
        self.drug_mapping = _load_json(drug_mapping_path) or {}
        self.condition_synonyms = {k: list(v) for k, v in _CONDITIONS}
        self.medical_stopwords = set(_KEEP_WORDS)

    def normalize_text(self, text: str, expand_abbreviations: bool = True) -> str:
        if not text:
            return text
        out = text
        if expand_abbreviations:
            # longest abbreviation first, whole words only
            for abbr, full in sorted(self.abbreviations.items(), key=lambda kv: len(kv[0]), reverse=True):
                out = _word(abbr).sub(full, out)
        low = out.lower()
        for generic, brands in self.drug_mapping.items():
            for brand in brands:
                if brand.lower() in low:
                    out = _word(brand).sub(f"{generic} ({brand})", out)
        for standard, synonyms in self.condition_synonyms.items():
            for syn in synonyms:
                if syn.lower() in out.lower():
                    out = _word(syn).sub(standard, out)
        return out

    def extract_medical_terms(self, text: str):
        low = text.lower()
        drugs = []
        for generic, brands in self.drug_mapping.items():
            if generic in low:
                drugs.append(generic)
            drugs.extend(f"{b} ({generic})" for b in brands if b.lower() in low)
        abbrs = [a for a in self.abbreviations if _word(a).search(text)]
        return {'drugs': drugs, 'conditions': [], 'abbreviations': abbrs}
