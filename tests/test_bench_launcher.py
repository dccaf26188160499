"""`bench.py --gpus N` started without a rank environment launches its own N rank processes and prints one line.

CPU: two ranks over gloo with the oracle-backed doubles (tests/bench_double.py) - the launcher, the shard build with
global BM25 statistics, the packed list exchange and the line's n_gpus.  GPU: the same launcher path with one rank over
RCCL (RAG_DIST_FORCE=1 makes a world of one take the process group and every exchange) and the real kernels.
"""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SMALL = ['--vocab', '400', '--dim', '64', '--embed-arch', 'tiny', '--ce-arch', 'tiny', '--no-extras', '--cpu-queries', '0',
         '--steps', '2', '--warmup', '1', '--rerank-top-n', '10']


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'RAG_DIST_FORCE')}
    env.update(extra)
    return env


def test_gpus_2_starts_two_ranks_and_prints_one_line():
    p = subprocess.run([sys.executable, os.path.join(HERE, 'bench_double.py'), '--gpus', '2', '--rows', '1500', '--batch', '6'] + SMALL,
                       capture_output=True, text=True, timeout=600, env=_clean_env())
    assert p.returncode == 0, p.stderr[-3000:]
    line = _line(p.stdout)
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['warmup'] == 1
    cfg = line['config']
    assert cfg['rows_total'] == 3000 and cfg['batch'] == 12 and cfg['batch_per_gpu'] == 6
    assert line['value'] > 0 and line['scaling'] == 'weak'
    assert '[rank 1] dense shard' in p.stderr                 # the second rank built its own shard


def test_a_failing_rank_fails_the_launcher():
    p = subprocess.run([sys.executable, os.path.join(HERE, 'bench_double.py'), '--gpus', '2', '--rows', '1500', '--batch', '6',
                        '--embed-arch', 'no-such-arch'] + SMALL[:4],
                       capture_output=True, text=True, timeout=600, env=_clean_env())
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith('{')]


def test_default_rows_make_eight_gpus_the_ten_million_row_configuration():
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module('bench')
    assert bench.default_rows(1) == 1_000_000
    assert bench.default_rows(8) * 8 == 10_000_000


@pytest.mark.gpu
def test_gpus_1_through_the_launcher_over_rccl():
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--rows', '20000', '--batch', '128',
                        '--no-extras', '--cpu-queries', '0', '--steps', '2', '--warmup', '1', '--embed-arch', 'minilm-l6',
                        '--dim', '384', '--vocab', '5000'],
                       capture_output=True, text=True, timeout=600, env=_clean_env(RAG_DIST_FORCE='1'))
    assert p.returncode == 0, p.stderr[-3000:]
    line = _line(p.stdout)
    assert line['n_gpus'] == 1 and line['rccl_ranks'] == 1
    assert line['config']['rows_total'] == 20000 and line['value'] > 0
    assert '[launcher]' in p.stderr
    # 128 queries x 50 pairs: the screened rerank ran, its two logit exchanges over RCCL included
    assert line['rerank_screen'] is not None and 0 < line['rerank_screen']['rescored_fraction'] <= 1
