"""CPU oracle for the embed -> search -> fuse -> rerank retrieval path.

TEST INFRASTRUCTURE ONLY.  Nothing under this package is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker.  The product code
(``domain-rag-system_amd/``) never imports it and raises when the HIP library
is missing instead of falling back to anything here.

What each module restates (reference paths are relative to the reference
checkout):

* ``tokenize_ref``  - ``src/indexing/bm25_indexer.py:50-78``
* ``bm25_ref``      - ``src/indexing/bm25_indexer.py:80-163`` plus the published
                      ``rank_bm25`` 0.2.x ``BM25Okapi`` algorithm it calls
* ``dense_ref``     - ``src/indexing/vector_indexer.py:105-230`` plus the
                      qdrant-client local-mode brute-force search it calls
* ``fusion_ref``    - ``src/indexing/hybrid_indexer.py:60-218``
* ``rerank_ref``    - ``src/reranker/rerank.py:29-68``

Pinning status
--------------
``fusion_ref``, ``tokenize_ref`` and ``rerank_ref`` are pinned: they are checked
against (a) fixtures produced by running the reference's own pure-Python
functions in the build container (``tests/golden/make_golden.py``) and (b) the
fused scores stored in the reference's ``results/*.json`` files, which decode
exactly as sums of ``1/(60+rank)``.

``bm25_ref`` and ``dense_ref`` restate third-party packages (``rank_bm25``,
``qdrant-client`` local mode) whose source is neither in the reference checkout
nor installed; the reference has no test or stored output that pins results at
those two boundaries.  For those two modules: PARITY UNPINNED (restatement of
the published upstream algorithm, checked only for self-consistency between a
literal per-document-dict form and the vectorised form used for timing).
"""
