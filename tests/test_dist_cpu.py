"""CPU: the row-sharded query path over gloo with world_size 2 gives the unsharded answer.
Kernels are replaced by oracle-backed doubles; what is under test is the exchange: global
BM25 statistics, per-shard lists with global row ids, all-gather + merge, replicated fusion,
cross-encoder pairs split across ranks, logits all-gather."""
import os
import socket
import subprocess
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run(world, out_path, fusion, metric='cosine', reranker='cross', device='cpu', env=None):
    port = _free_port()
    full_env = dict(os.environ, **(env or {}))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, 'dist_worker.py'), str(r), str(world), str(port),
                               out_path, fusion, metric, reranker, device], env=full_env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    return torch.load(out_path, weights_only=True)


@pytest.mark.parametrize('fusion', ['rrf', 'weighted'])
def test_two_shards_equal_one(tmp_path, fusion):
    one = _run(1, str(tmp_path / 'one.pt'), fusion)
    two = _run(2, str(tmp_path / 'two.pt'), fusion)
    assert torch.equal(one['fused_row'], two['fused_row'])
    assert torch.equal(one['fused_score'], two['fused_score'])          # fp64, bit for bit
    assert torch.equal(one['fused_count'], two['fused_count'])
    assert torch.equal(one['rows'], two['rows']) and torch.equal(one['count'], two['count'])
    assert torch.allclose(one['rerank_score'], two['rerank_score'], atol=1e-5)
    assert one['rows'].shape == (9, 4) and int(one['count'].min()) == 4


def test_two_shards_equal_one_with_a_euclidean_index(tmp_path):
    """Distances are smaller-is-better: the shard merge must keep each query's k SMALLEST distances."""
    one = _run(1, str(tmp_path / 'one.pt'), 'rrf', 'euclidean')
    two = _run(2, str(tmp_path / 'two.pt'), 'rrf', 'euclidean')
    assert torch.equal(one['fused_row'], two['fused_row'])
    assert torch.equal(one['fused_score'], two['fused_score'])
    assert torch.equal(one['rows'], two['rows']) and torch.equal(one['count'], two['count'])
    cos = _run(1, str(tmp_path / 'cos.pt'), 'rrf', 'cosine')
    assert one['rows'].shape == cos['rows'].shape        # (unit vectors: both metrics rank alike, both paths run)


def test_two_shards_equal_one_with_the_simple_reranker(tmp_path):
    """Simple reranker from resident rows: each rank scores the pool rows it holds, the scores are summed."""
    one = _run(1, str(tmp_path / 'one.pt'), 'rrf', 'cosine', 'simple')
    two = _run(2, str(tmp_path / 'two.pt'), 'rrf', 'cosine', 'simple')
    assert torch.equal(one['fused_row'], two['fused_row'])
    assert torch.equal(one['rows'], two['rows']) and torch.equal(one['count'], two['count'])
    assert torch.equal(one['rerank_score'], two['rerank_score'])          # the same float32 dot products


def test_two_shards_with_the_screened_rerank_equal_one_without(tmp_path):
    """Screen-then-rescore across ranks: the cheap logits are gathered, every rank takes the same decisions, the chosen
    pairs are split across the ranks and their exact logits gathered.  Same final rows and scores as one rank that
    scores every pair exactly."""
    one = _run(1, str(tmp_path / 'one.pt'), 'rrf', 'cosine', 'cross')
    two = _run(2, str(tmp_path / 'two.pt'), 'rrf', 'cosine', 'cross-screened')
    assert torch.equal(one['fused_row'], two['fused_row']) and torch.equal(one['fused_score'], two['fused_score'])
    assert torch.equal(one['rows'], two['rows']) and torch.equal(one['count'], two['count'])
    assert torch.allclose(one['rerank_score'], two['rerank_score'], atol=1e-5)
    assert 0.0 < float(two['rescored_fraction']) <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize('reranker', ['cross', 'simple'])
def test_two_ranks_on_one_gpu_equal_one_with_the_real_kernels(tmp_path, reranker):
    """The same comparison with librag_hip's kernels: two processes share GPU 0, each holds half of the rows,
    the exchanges (query-embedding slices, per-shard lists, logits / rescored rows) go over gloo."""
    one = _run(1, str(tmp_path / 'one.pt'), 'rrf', 'cosine', reranker, 'cuda')
    two = _run(2, str(tmp_path / 'two.pt'), 'rrf', 'cosine', reranker, 'cuda')
    assert torch.equal(one['fused_row'], two['fused_row'])
    assert torch.equal(one['fused_score'], two['fused_score'])
    assert torch.equal(one['rows'], two['rows']) and torch.equal(one['count'], two['count'])
    assert torch.allclose(one['rerank_score'], two['rerank_score'], atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('reranker', ['cross', 'simple'])
def test_single_rank_over_rccl_takes_every_exchange(tmp_path, reranker):
    """The RCCL branch of the exchanges (device tensors straight into the collectives) with one rank: RAG_DIST_FORCE=1
    makes a world of one take every all-gather / all-reduce and merge (each an identity), on backend "nccl".  Same
    answer as the run without a process group."""
    one = _run(1, str(tmp_path / 'one.pt'), 'rrf', 'cosine', reranker, 'cuda')
    forced = _run(1, str(tmp_path / 'forced.pt'), 'rrf', 'cosine', reranker, 'cuda', env={'RAG_DIST_FORCE': '1'})
    assert torch.equal(one['fused_row'], forced['fused_row']) and torch.equal(one['fused_score'], forced['fused_score'])
    assert torch.equal(one['rows'], forced['rows']) and torch.equal(one['count'], forced['count'])
    assert torch.allclose(one['rerank_score'], forced['rerank_score'], atol=1e-5)


@pytest.mark.gpu
def test_two_ranks_with_the_packed_cross_encoder_and_the_screen(tmp_path):
    """The unpadded (packed) cross-encoder route of the real kernels across two ranks, 99 pairs (an odd count: the last
    rank's share is one shorter than the padded piece that is gathered), with and without the float16 screen."""
    one = _run(1, str(tmp_path / 'one.pt'), 'rrf', 'cosine', 'cross-native', 'cuda')
    two = _run(2, str(tmp_path / 'two.pt'), 'rrf', 'cosine', 'cross-native', 'cuda')
    scr = _run(2, str(tmp_path / 'scr.pt'), 'rrf', 'cosine', 'cross-native-screened', 'cuda')
    for other in (two, scr):
        assert torch.equal(one['fused_row'], other['fused_row']) and torch.equal(one['fused_score'], other['fused_score'])
        assert torch.equal(one['rows'], other['rows']) and torch.equal(one['count'], other['count'])
        assert torch.allclose(one['rerank_score'], other['rerank_score'], atol=2e-6)
    assert 0.0 < float(scr['rescored_fraction']) <= 1.0
