"""Oracle: the BM25 tokenizer (test infrastructure, never imported by the product).

Restates ``BM25Indexer.tokenize`` (reference ``src/indexing/bm25_indexer.py:50-78``):
lower-case, ``re.findall(r'\\b\\w+\\b')`` with Unicode ``\\w``, keep a token when it
has at least two characters or when it equals a lower-cased abbreviation key that
occurs as a substring of the lower-cased text.  The abbreviation keys are the
default table of ``MedicalTermNormalizer._load_abbreviations`` (reference
``src/preprocessing/medical_term_normalizer.py:170-218``).

Pinned by ``tests/golden/reference_pure_python.json`` (outputs of the reference
function itself).
"""
import re

# keys of the reference's default abbreviation table, in its order
ABBREVIATION_KEYS = (
    'MI', 'CHF', 'HTN', 'CAD', 'AF', 'CVA',
    'DM', 'T1DM', 'T2DM', 'HbA1c',
    'COPD', 'URI', 'SOB',
    'GERD', 'IBD', 'IBS',
    'Rx', 'Sx', 'Tx', 'Dx', 'Hx', 'pt', 'w/', 'w/o',
    'WBC', 'RBC', 'BP', 'HR', 'BUN',
    'NSAID', 'ACE', 'ARB', 'SSRI', 'PPI',
)

_WORD = re.compile(r'\b\w+\b')


def tokenize_ref(text, abbreviation_keys=ABBREVIATION_KEYS):
    low = text.lower()
    kept_short = set()
    for key in abbreviation_keys:
        k = key.lower()
        if k in low:
            kept_short.add(k)
    out = []
    for tok in _WORD.findall(low):
        if len(tok) >= 2 or tok in kept_short:
            out.append(tok)
    return out
