"""bench.py on the CPU with the oracle-backed doubles in place of the GPU entry points (TEST INFRASTRUCTURE).

Runs the script's host side only - the rank launcher, the shard build with global BM25 statistics, the exchanges over
gloo, the JSON line - so that `bench.py --gpus N` can be exercised in a container without a GPU.  The launcher starts
its ranks from RAG_BENCH_ENTRY, i.e. from this file, so every rank installs the doubles as well.  Nothing this prints is a
measurement.

    python tests/bench_double.py --gpus 2 --rows 1500 --batch 6 ...
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'domain-rag-system_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

if __name__ == '__main__':
    os.environ['RAG_BENCH_DEVICE'] = 'cpu'
    os.environ['RAG_BENCH_ENTRY'] = os.path.abspath(__file__)
    from tests import fake_backend
    fake_backend.install()
    runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
