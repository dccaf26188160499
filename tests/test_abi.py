"""CPU: the C-ABI library loads and exports every symbol include/rag_hip.h declares, with a
ctypes prototype for each (no compute call is made)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'rag_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rag_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    import rag_hip
    lib = rag_hip.load_library()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f'{n} is declared in rag_hip.h but not exported'
    assert sorted(rag_hip.SIGNATURES) == names, 'ctypes prototypes and header disagree'
    assert lib.rag_version() >= 100


def test_error_string_is_reachable_without_a_gpu():
    import rag_hip
    lib = rag_hip.load_library()
    assert isinstance(rag_hip.last_error(), str)
    # argument validation happens before any device call
    rc = lib.rag_topk_segmented(None, None, 1, 0, None, None)
    assert rc == -5 and 'top_k' in rag_hip.last_error()
    rc = lib.rag_dense_search(None, None, 1, 5, None, None, None, None)
    assert rc == -1 and 'null handle' in rag_hip.last_error()


def test_header_cites_the_reference_for_each_group():
    text = open(os.path.join(ROOT, 'include', 'rag_hip.h')).read()
    for cite in ('src/indexing/vector_indexer.py:202-209', 'src/indexing/bm25_indexer.py:127-128',
                 'src/indexing/hybrid_indexer.py:112-156', 'src/reranker/rerank.py:42-43',
                 'src/reranker/rerank.py:79-84'):
        assert cite in text
