#!/usr/bin/env python3
"""End-to-end throughput of the hybrid retrieval path on MI355X.

One step = one batch of synthetic queries through
embed -> dense kNN -> BM25 -> RRF -> cross-encoder rerank -> top-k, everything resident
in HBM when the timed region starts.  Prints ONE JSON line (rank 0).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N --steps K --warmup W         (starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

With N > 1 the corpus rows are sharded (each rank holds --rows rows: weak scaling in corpus
size; the default is 1.25 M rows per rank, so that N = 8 is the 10 M-row configuration, and
1 M rows at N = 1) and a step is one batch of --batch x N queries: every rank searches its rows
for all of them, the per-shard top-k lists are all-gathered over RCCL and merged, and the
cross-encoder pairs are split across the ranks (--batch x rerank_top_n pairs per GPU whatever N
is).  Started without a rank environment and with --gpus N > 1, the script launches N fresh rank
processes of itself (torch.distributed.run) before it makes any GPU call, hands rank 0's JSON
line through on stdout and exits with the launcher's status.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'domain-rag-system_amd'))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X dense fp32 matrix peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 matrix peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=1024,
                    help='queries per step and GPU: a step is one batch of batch x N queries (every rank searches its rows for all\n'
                         'of them and reranks 1/N of the pairs, so the rerank work per GPU stays fixed as N grows)')
    ap.add_argument('--rows', type=int, default=None,
                    help='corpus rows per GPU (default: 1,000,000 on one GPU, 1,250,000 per GPU on several: 10 M rows on eight)')
    ap.add_argument('--dim', type=int, default=768)
    ap.add_argument('--vocab', type=int, default=200_000)
    ap.add_argument('--top-k', type=int, default=5)
    ap.add_argument('--rerank-top-n', type=int, default=50)
    ap.add_argument('--ce-dtype', default='fp32', choices=['fp32', 'bf16', 'fp32-torch'],
                    help="cross-encoder arithmetic: 'fp32' = float32 weights and activations, GEMMs and attention as "
                         "split-fp16 matrix-core products with fp32 accumulation (the reference's precision, default); "
                         "'bf16' = the bfloat16 kernels (faster, does not hold the fp32 order); 'fp32-torch' = plain PyTorch ops")
    ap.add_argument('--ce-screen', default='fp16', choices=['fp16', 'none'],
                    help="with --ce-dtype fp32: 'fp16' = every pair is first scored by a float16 copy of the cross-encoder (one "
                         "matrix-core product per product), and only the pairs within a self-checked margin of their query's "
                         "k-th best go through the float32-equivalent forward, from which the final order and scores come; "
                         "'none' = every pair through the float32-equivalent forward")
    ap.add_argument('--screen-one-stage', action='store_true',
                    help='tuning aid: the screen with one rescoring forward and a two-sided margin (2 x safety x error)')
    ap.add_argument('--screen-safety', type=float, default=1.5,
                    help='margin of the screen = (one stage: 2 x) this x the largest within-query difference between cheap and exact logits '
                         'measured on the rescored pairs')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the secondary measurements (rerank_top_n=30, bf16 cross-encoder, batch sweep, host-buffer run) '
                         'and the precision check: for profiler runs')
    ap.add_argument('--extra-steps', type=int, default=4, help='timed steps of each secondary measurement')
    ap.add_argument('--ce-pairs-per-batch', type=int, default=16384,
                    help='pairs per cross-encoder forward (an unpadded token stream when the model runs the librag_hip kernels)')
    ap.add_argument('--tune-gemms', default=None, metavar='FILE',
                    help='measure hipBLASLt solutions for the encoder GEMM shapes of this workload and write the '
                         'table to FILE (slow; the default run only reads the table shipped with the package)')
    ap.add_argument('--no-gemm-table', action='store_true', help='ignore the shipped hipBLASLt solution table')
    ap.add_argument('--embed-torch-ops', action='store_true',
                    help='run the embedder on padded batches with PyTorch ops instead of the unpadded librag_hip path')
    ap.add_argument('--embed-fp32-gemm', action='store_true',
                    help='run the embedder GEMMs as fp32 GEMMs instead of split-bf16 products with fp32 accumulation')
    ap.add_argument('--ce-fused-attn-out', action='store_true',
                    help='tuning aid: attention-output projection + LayerNorm inside the feed-forward kernel')
    ap.add_argument('--ce-fused-norm-mlp', action='store_true',
                    help='tuning aid: residual + LayerNorm after the output projection inside the feed-forward kernel '
                         '(the default of the float16 screening forward)')
    ap.add_argument('--ce-no-fused-norm-mlp', action='store_true', help='tuning aid: ... and as a pass of its own there too')
    ap.add_argument('--ce-no-fused-qkv-attention', action='store_true',
                    help='tuning aid: q|k|v projection and self-attention of the 16-bit forward as two kernels')
    ap.add_argument('--ce-fused-attn-norm', action='store_true',
                    help='tuning aid: residual + LayerNorm after the output projection inside rag_enc_linear384_dev')
    ap.add_argument('--pipeline', action='store_true',
                    help='overlap the retrieval half of batch i+1 with the rerank of batch i on a second stream '
                         '(BatchSearcher.search_many; measured: no gain, the rerank kernels already fill the chip)')
    ap.add_argument('--ce-library-linear', action='store_true',
                    help='tuning aid: q|k|v and output projections as library GEMMs instead of rag_enc_linear384_dev')
    ap.add_argument('--ce-library-mlp', action='store_true',
                    help='tuning aid: feed-forward blocks as library GEMMs + tail kernels instead of rag_enc_mlp_dev')
    ap.add_argument('--ce-torch-ops', action='store_true',
                    help='run attention and LayerNorm of the cross-encoder as PyTorch ops instead of the librag_hip kernels')
    ap.add_argument('--embed-arch', default='bert-base')
    ap.add_argument('--ce-arch', default='minilm-l6')
    ap.add_argument('--cpu-queries', type=int, default=32,
                    help='queries timed on the host for cpu_baseline (0 = skip); 32 are 10-15 s of work on 16 cores')
    ap.add_argument('--stage-times', action='store_true', help='extra untimed pass with per-stage events')
    args = ap.parse_args()
    return args


def default_rows(world):
    """Rows per GPU when --rows is not given: the 1M-row configuration on one GPU, 1.25 M per GPU on several
    (eight GPUs hold the 10M-row configuration)."""
    return 1_000_000 if world == 1 else 1_250_000


def launch_ranks(args):
    """`bench.py --gpus N` started by hand (no rank environment): run N fresh rank processes of this script under
    torch.distributed.run and pass rank 0's line through.  Called before this process has made any GPU call (it never
    makes one: the ranks are children, and this process only waits for them)."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % args.gpus,
           '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.environ.get('RAG_BENCH_ENTRY', os.path.abspath(__file__))] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, min(16, (os.cpu_count() or 1) // max(args.gpus, 1)))))
    log('[launcher] ' + ' '.join(cmd))
    return subprocess.call(cmd, env=env)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def token_id_limit(args):
    """Synthetic token ids are drawn below the smaller of the two models' vocabularies (30522 for the named ones)."""
    from indexing import bert
    return min(bert.ARCHS[getattr(args, 'embed_arch', 'bert-base')].vocab_size,
               bert.ARCHS[getattr(args, 'ce_arch', 'minilm-l6')].vocab_size)


def build_shard(args, rank, world, dev, dist):
    """Synthetic shard of `args.rows` rows: embeddings, postings, chunk tokens, canon."""
    import rag_hip
    from indexing.bm25_indexer import build_arrays_device
    from tests import synth
    n, dim, v = args.rows, args.dim, args.vocab
    row_base = rank * n
    t0 = time.time()
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    rows = torch.empty((n, dim), dtype=torch.float32, device=dev)
    step = 1 << 18
    for s in range(0, n, step):
        blk = torch.randn((min(step, n - s), dim), generator=g, device=dev, dtype=torch.float32)
        rows[s:s + step] = torch.nn.functional.normalize(blk, dim=1)
    dense = rag_hip.DenseIndex(rows, metric='cosine', row_base=row_base, borrow=True)
    log(f'[rank {rank}] dense shard {n}x{dim} in {time.time() - t0:.1f}s')

    t0 = time.time()
    term, doc, ln = synth.sparse_corpus(n, v, seed=2000 + rank)
    stats = None
    if dist is not None:
        df = np.bincount(np.unique(term * n + doc) // n, minlength=v).astype(np.int64)
        tdf = torch.from_numpy(df).to(dev)
        tot = torch.tensor([int(ln.sum()), n], dtype=torch.int64, device=dev)
        dist.all_reduce(tdf)
        dist.all_reduce(tot)
        stats = dict(df=tdf.cpu().numpy(), total_len=int(tot[0]), n_docs=int(tot[1]))
    arr = build_arrays_device(term, doc, ln, v, 1.5, 0.75, global_stats=stats, device=dev)
    bm25 = rag_hip.Bm25Index(arr.indptr, arr.rows, arr.impacts(), arr.idf, n, row_base=row_base)
    log(f'[rank {rank}] bm25 shard: {arr.rows.shape[0]} postings in {time.time() - t0:.1f}s')

    # chunk tokens for the cross-encoder: doc length + 2 synthetic ids per chunk
    clen = torch.from_numpy(ln + 2).to(dev)
    off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(clen, 0)
    tok = torch.randint(1000, token_id_limit(args), (int(off[-1]),), generator=g, device=dev, dtype=torch.int32)
    if dist is not None:                # replicate the token store (small) so any rank can pack any pair
        lens = [torch.empty_like(clen) for _ in range(world)]
        dist.all_gather(lens, clen)
        sizes = [int(l.sum()) for l in lens]
        longest = max(sizes)            # shards differ in length: gather equal-sized (padded) pieces
        mine = torch.zeros((longest,), dtype=torch.int32, device=dev)
        mine[:tok.shape[0]] = tok
        toks = [torch.empty((longest,), dtype=torch.int32, device=dev) for _ in range(world)]
        dist.all_gather(toks, mine)
        tok = torch.cat([t[:s] for t, s in zip(toks, sizes)])
        clen = torch.cat(lens)
        off = torch.zeros(n * world + 1, dtype=torch.int64, device=dev)
        off[1:] = torch.cumsum(clen, 0)
    # 0.5 % of the rows share a chunk_id with an earlier row (same on every rank)
    n_all = n * world
    rng = np.random.default_rng(7)
    canon = np.arange(n_all, dtype=np.uint32)
    dup = np.sort(rng.choice(np.arange(1, n_all), size=max(1, n_all // 200), replace=False))
    for d, u in zip(dup.tolist(), rng.random(dup.shape[0]).tolist()):
        canon[d] = canon[int(u * d)]    # the canonical row of an earlier row
    host = dict(term=term, doc=doc, ln=ln, arr=arr, canon=canon)
    return rows, dense, bm25, tok, off, torch.from_numpy(canon).to(dev), host


def make_queries(args, n_batches, dev, seed=99):
    """Same queries on every rank: embedder token ids, BM25 term ids, cross-encoder token ids."""
    rng = np.random.default_rng(seed)
    from tests import synth
    cdf = np.cumsum(synth.zipf_probs(args.vocab))
    out = []
    for _ in range(n_batches):
        b = args.batch
        qlen = rng.integers(8, 25, size=b)
        smax = int(qlen.max()) + 2
        ids = np.zeros((b, smax), np.int64)
        mask = np.zeros((b, smax), np.int64)
        ce_tok, ce_off = [], [0]
        for i in range(b):
            body = rng.integers(1000, token_id_limit(args), size=int(qlen[i]))
            ids[i, 0], ids[i, 1:1 + qlen[i]], ids[i, 1 + qlen[i]] = 101, body, 102
            mask[i, :qlen[i] + 2] = 1
            ce_tok.append(body)
            ce_off.append(ce_off[-1] + int(qlen[i]))
        nt = rng.integers(3, 13, size=b)
        terms = np.minimum(np.searchsorted(cdf, rng.random(int(nt.sum()))), args.vocab - 1).astype(np.int32)
        oov = rng.random(b) < 0.01
        qoff = np.zeros(b + 1, np.int32)
        qoff[1:] = np.cumsum(nt)
        for i in np.nonzero(oov)[0]:
            terms[qoff[i]:qoff[i + 1]] = -1
        out.append(dict(
            ids=torch.from_numpy(ids).to(dev), mask=torch.from_numpy(mask).to(dev),
            qterms=torch.from_numpy(terms).to(dev), qoff=torch.from_numpy(qoff).to(dev), n_terms=int(nt.sum()),
            ce_tok=torch.from_numpy(np.concatenate(ce_tok).astype(np.int32)).to(dev),
            ce_off=torch.from_numpy(np.asarray(ce_off, np.int64)).to(dev),
            host=dict(ids=ids, mask=mask, terms=terms, qoff=qoff, ce_tok=ce_tok)))
    return out


def cpu_baseline(args, searcher_models, rows_dev, host, queries, gpu_out):
    """The oracle (CPU restatement of the reference's algorithms) on a few queries of the same
    workload, timed on the host cores.  Also compares its ids with the GPU path's."""
    from oracle.bm25_ref import BM25OkapiArrays, rank_rows_ref
    from oracle.dense_ref import dense_topk_ref
    from oracle.fusion_ref import rrf_keys
    from oracle.rerank_ref import rerank_order_ref
    from indexing import bert
    nq = args.cpu_queries
    cores = os.cpu_count() if os.environ.get('OMP_NUM_THREADS') is None else int(os.environ['OMP_NUM_THREADS'])
    torch.set_num_threads(cores)
    bi_cpu, ce_cpu = searcher_models
    rows = rows_dev.cpu().numpy()
    t_build = time.time()
    ref = BM25OkapiArrays.from_token_ids(host['term'], host['doc'], host['ln'], args.vocab)
    t_build = time.time() - t_build
    canon = host['canon']
    q = queries['host']
    t = args.rerank_top_n
    k_list = 2 * t
    agree_fused = agree_final = 0
    t0 = time.time()
    for i in range(nq):
        n_tok = int(q['mask'][i].sum())
        with torch.no_grad():
            v = bi_cpu(torch.from_numpy(q['ids'][i:i + 1, :n_tok]), None, torch.from_numpy(q['mask'][i:i + 1, :n_tok]),
                       normalize=True).numpy()[0]
        d_rows, _ = dense_topk_ref(rows, v, k_list)
        terms = q['terms'][q['qoff'][i]:q['qoff'][i + 1]].tolist()
        s = ref.get_scores_ids(terms)
        b_rows = rank_rows_ref(s, k_list)
        fused = rrf_keys([int(canon[r]) for r in d_rows], [int(canon[r]) for r in b_rows], top_k=t)
        both = list(d_rows) + list(b_rows)
        f_rows = [int(both[p]) for p, _ in fused]
        # cross-encoder on the host: pairs of (query tokens, chunk tokens)
        c_tok = gpu_out['chunk_tokens'](f_rows)
        qt = list(q['ce_tok'][i])
        ids_l = [[101] + qt + [102] + c + [102] for c in c_tok]
        typ_l = [[0] * (len(qt) + 2) + [1] * (len(c) + 1) for c in c_tok]
        logits = []
        for s0 in range(0, len(ids_l), 32):
            ids_t, m_t = bert.pad_batch(ids_l[s0:s0 + 32], 0, 'cpu')
            ty_t, _ = bert.pad_batch(typ_l[s0:s0 + 32], 0, 'cpu')
            with torch.no_grad():
                logits.extend(ce_cpu(ids_t, ty_t, m_t).tolist())
        order = rerank_order_ref(logits, args.top_k)
        final = [f_rows[j] for j in order]
        agree_fused += int(f_rows == gpu_out['fused_row'][i][:len(f_rows)].tolist())
        agree_final += int(final == gpu_out['rows'][i][:len(final)].tolist())
    dt = time.time() - t0
    return {'value': nq / dt, 'unit': 'queries/sec', 'cores': cores, 'kind': 'port',
            'sample': f'{nq} queries of the same workload (full {args.rows}-row corpus), single process, '
                      f'oracle dense+BM25+RRF in numpy/Python, BERT forwards in PyTorch CPU fp32; '
                      f'index build ({t_build:.1f}s) not included',
            'fused_ids_equal_gpu': f'{agree_fused}/{nq}', 'top_k_ids_equal_gpu': f'{agree_final}/{nq}'}


def main():
    args = parse()
    if 'RANK' not in os.environ and (args.gpus > 1 or os.environ.get('RAG_DIST_FORCE') == '1'):
        sys.exit(launch_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        log(f'[bench] --gpus {args.gpus} but the launcher started {world} rank(s): running {world}')
    if args.rows is None:
        args.rows = default_rows(world)
    batch_per_gpu = args.batch
    args.batch = batch_per_gpu * world             # the step's batch: every rank sees all of its queries
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    # RAG_BENCH_DEVICE=cpu: the host side of this script (launcher, shard build with global statistics, exchanges, the
    # line) on the CPU over gloo.  The GPU entry points of rag_hip must have been replaced by the caller (the CPU tests
    # do that with their doubles, tests/bench_double.py); nothing this mode prints is a measurement.
    on_cpu = os.environ.get('RAG_BENCH_DEVICE') == 'cpu'
    # rehearsal of the multi-rank code path on a one-GPU box: every rank on device 0, gloo instead of RCCL
    rehearse = world > 1 and os.environ.get('RAG_BENCH_REHEARSE') == '1'
    if rehearse:
        local = 0
        import faulthandler                      # a rank that waits forever shows where
        faulthandler.dump_traceback_later(100, repeat=True)
    # RAG_DIST_FORCE=1: the process group and every exchange also with a single rank (launched through
    # torch.distributed.run --nproc-per-node 1): drives the RCCL code path on a one-GPU box
    force = os.environ.get('RAG_DIST_FORCE') == '1' and 'RANK' in os.environ
    if world > 1 or force:
        import torch.distributed as dist
        if not on_cpu:
            torch.cuda.set_device(local)
        if rehearse or on_cpu:
            dist.init_process_group('gloo')
        else:
            # RCCL prints a version banner on stdout when its first communicator comes up: keep stdout for the one
            # JSON line (file descriptor 1 points at stderr until the communicator exists)
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group('nccl', device_id=torch.device('cuda', local))
                dist.barrier()
                torch.cuda.synchronize()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
    dev = torch.device('cpu') if on_cpu else torch.device('cuda', local)
    import rag_hip
    if not on_cpu:
        torch.cuda.set_device(dev)
        rag_hip.init(local)
    sync = (lambda: None) if on_cpu else torch.cuda.synchronize
    from indexing import bert
    from indexing.batch import BatchSearcher, TokenStore

    from indexing import gemm_tuning
    tuned = False
    if on_cpu:
        pass
    elif args.tune_gemms:
        tuned = gemm_tuning.enable(tune=True, table=args.tune_gemms)
    elif not args.no_gemm_table:
        tuned = gemm_tuning.enable(tune=False)
    log(f'[rank {rank}] hipBLASLt solution table: ' + ('tuning into ' + args.tune_gemms if args.tune_gemms
                                                      else 'in use' if tuned else 'not used'))
    rows, dense, bm25, tok, off, canon, host = build_shard(args, rank, world, dev, dist)
    bi, _, _ = bert.load_model('random:' + args.embed_arch, 'bi', dev, native=not (args.embed_torch_ops or on_cpu),
                               split_gemm=not (args.embed_fp32_gemm or on_cpu))
    def load_ce(kind):
        """The cross-encoder in one of the three arithmetic forms of --ce-dtype."""
        if on_cpu:
            kind = 'fp32-torch'
        if kind in ('bf16', 'fp16'):
            m, _, _ = bert.load_model('random:' + args.ce_arch, 'cross', dev,
                                      dtype=torch.bfloat16 if kind == 'bf16' else torch.float16,
                                      native=not args.ce_torch_ops)
        elif kind == 'fp32':
            m, _, _ = bert.load_model('random:' + args.ce_arch, 'cross', dev, dtype=torch.float32, native=True,
                                      split_gemm=True, split_dtype=torch.float16)
        else:
            m, _, _ = bert.load_model('random:' + args.ce_arch, 'cross', dev, dtype=torch.float32)
        m.bert.fused_mlp = not args.ce_library_mlp
        m.bert.native_linear = not args.ce_library_linear
        m.bert.fused_attn_norm = bool(args.ce_fused_attn_norm)
        m.bert.fused_tail = bool(args.ce_fused_attn_out)
        m.bert.fused_norm_mlp = bool(args.ce_fused_norm_mlp) or (kind == 'fp16' and not args.ce_no_fused_norm_mlp)
        m.bert.fused_qkv_attention = not getattr(args, 'ce_no_fused_qkv_attention', False)
        return m

    def make_searcher(model, top_n, with_dist=True, screen=None):
        return BatchSearcher(dense, bm25, canon=canon, bi_encoder=bi, cross_encoder=model, screen_encoder=screen,
                             screen_safety=args.screen_safety, screen_two_stage=not args.screen_one_stage,
                             chunk_tokens=TokenStore(tok, off, 0), top_k=args.top_k, rerank_top_n=top_n,
                             ce_pairs_per_batch=args.ce_pairs_per_batch, dist=dist if with_dist else None)

    ce = load_ce(args.ce_dtype)
    ce_screen = load_ce('fp16') if (args.ce_dtype == 'fp32' and args.ce_screen == 'fp16' and not on_cpu) else None
    searcher = make_searcher(ce, args.rerank_top_n, screen=ce_screen)
    n_batches = args.steps + args.warmup
    queries = make_queries(args, max(n_batches, 1), dev)
    # work buffers of the two indexes sized once, before anything is timed: no search of the run allocates
    searcher.reserve(args.batch, max(int(q['n_terms']) for q in queries))
    if args.tune_gemms and searcher.ce_packed:
        gemm_tuning.tune_cross_encoder(ce, searcher.ce_tokens_per_batch, searcher.ce_token_bucket)

    def step(q):
        return searcher.search(q['ids'], q['mask'], q['qterms'], q['qoff'], q['n_terms'], q['ce_tok'], q['ce_off'])

    def barrier():
        if dist is not None:
            dist.barrier()
        sync()

    def as_args(q):
        return (q['ids'], q['mask'], q['qterms'], q['qoff'], q['n_terms'], q['ce_tok'], q['ce_off'])

    def run(batch_list):
        # K steps = K batches through the whole path, one after the other; with --pipeline the retrieval half
        # of batch i+1 overlaps the rerank of batch i on a second stream (all K x both halves inside the timed region)
        if not args.pipeline:
            return [step(q) for q in batch_list]
        return searcher.search_many([as_args(q) for q in batch_list])

    run(queries[:args.warmup])
    barrier()
    screen_before = dict(searcher.screen_stats)
    if not on_cpu:
        dense.profile(True)
        rag_hip.enc_mlp_profile(True)
        rag_hip.enc_linear_ln_split_profile(True)
    t0 = time.perf_counter()
    out = run(queries[args.warmup:args.warmup + args.steps])[-1]
    barrier()
    dt = time.perf_counter() - t0
    screen_after = dict(searcher.screen_stats)
    scan_ms = mlp_ms = lls_ms = np.zeros(0, np.float32)
    mlp_rows = lls_rows = lls_k = np.zeros(0, np.int64)
    if not on_cpu:
        scan_ms = dense.profile_read()
        dense.profile(False)
        mlp_ms, mlp_rows = rag_hip.enc_mlp_profile_read()
        rag_hip.enc_mlp_profile(False)
        lls_ms, lls_rows, lls_k = rag_hip.enc_linear_ln_split_profile_read()
        rag_hip.enc_linear_ln_split_profile(False)
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0])

    # per-stage breakdown (untimed extra pass with a sync after every stage)
    stage_ms = {}
    if not on_cpu:
        searcher.timing = True
        step(queries[0])
        torch.cuda.synchronize()
        searcher.timing = False
        stage_ms = {k: round(v, 3) for k, v in searcher.times.as_dict().items()}

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    n_total = args.rows * world
    t = args.rerank_top_n
    # the dense scan.  Algorithmic work of one launch: FLOPs = 2 * rows * dim * batch, bytes = 4 * rows * dim
    # (one pass over the shard).  The split form runs three bf16 products per algorithmic one, so it is
    # priced against the bf16 matrix peak and its executed rate is given beside the algorithmic one; the
    # fp32 form is priced against the fp32 matrix peak.
    roofline = None
    if len(scan_ms):
        flops = 2.0 * args.rows * args.dim * args.batch
        scan_s = float(np.mean(scan_ms)) / 1e3
        achieved = flops / scan_s / 1e12
        dense_split = bool(getattr(dense, 'split', False))
        dense_peak = MFMA_BF16_PEAK_TFLOPS if dense_split else MFMA_F32_PEAK_TFLOPS
        roofline = {'kernel': '%s (%s)' % ('dense_scan_dma_kernel' if dense_split and args.batch >= 256 else 'dense_scan_kernel',
                                           'split-bf16 products, fp32 accumulation' if dense_split else 'fp32'),
                    'bound': 'mfma', 'achieved': round(achieved, 2),
                    'peak': dense_peak, 'unit': 'TFLOP/s', 'frac': round(achieved / dense_peak, 4),
                    'traffic': None, 'avg_launch_ms': round(scan_s * 1e3, 3),
                    'algorithmic_gbs': round(4.0 * args.rows * args.dim / scan_s / 1e9, 1)}
        if dense_split:
            roofline['executed_tflops'] = round(3 * achieved, 1)
            roofline['executed_frac'] = round(3 * achieved / dense_peak, 4)
            roofline['vs_fp32_matrix_peak'] = round(achieved / MFMA_F32_PEAK_TFLOPS, 3)
        # memory-side bytes of one launch from the PMC passes (FETCH_SIZE doubled as gfx950 needs,
        # plus WRITE_SIZE), measured with scripts/kernel_bench.py on this workload and kept with
        # the profiles; null when the workload differs from the one that was measured
        pmc_path = os.path.join(ROOT, 'profiles', 'pmc_dense_scan.json')
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pmc = json.load(f)
            w = pmc.get('workload', {})
            if (w.get('rows'), w.get('dim'), w.get('batch'), w.get('k')) == (args.rows, args.dim, args.batch, 2 * t):
                roofline['traffic'] = round(float(pmc['traffic_bytes_per_launch']) / 1e9, 2)
                roofline['traffic_unit'] = 'GB per launch (rocprofv3 --pmc, separate passes)'

    # the kernel with the largest share of a step: the cross-encoder's feed-forward block.  Algorithmic
    # work of a launch: 4 * rows * hidden * intermediate FLOPs (two products), 2 * rows * hidden * 2 bytes.
    ce_cfg = ce.bert.config
    roofline_ff = None
    if len(mlp_ms):
        ff_flops = 4.0 * float(mlp_rows.sum()) * ce_cfg.hidden_size * ce_cfg.intermediate_size
        ff_s = float(mlp_ms.sum()) / 1e3
        roofline_ff = {'kernel': 'mlp_fused_kernel (feed-forward block of a layer in one kernel; %s)' % (
                           'float16 instance, the screening pass' if ce_screen is not None else 'bfloat16'),
                       'bound': 'mfma', 'achieved': round(ff_flops / ff_s / 1e12, 1),
                       'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                       'frac': round(ff_flops / ff_s / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), 'traffic': None,
                       'launches_per_step': len(mlp_ms) // max(args.steps, 1),
                       'avg_launch_ms': round(float(mlp_ms.mean()), 3), 'avg_launch_rows': int(mlp_rows.mean()),
                       'ms_per_step': round(float(mlp_ms.sum()) / max(args.steps, 1), 2),
                       'algorithmic_gbs': round(4.0 * float(mlp_rows.sum()) * ce_cfg.hidden_size / ff_s / 1e9, 1)}
        ff_pmc = os.path.join(ROOT, 'profiles', 'pmc_mlp.json')
        if os.path.exists(ff_pmc):
            with open(ff_pmc) as f:
                per_row = float(json.load(f)['traffic_bytes_per_row'])
            roofline_ff['traffic'] = round(per_row * float(mlp_rows.mean()) / 1e9, 2)
            roofline_ff['traffic_unit'] = 'GB per launch (bytes per row from rocprofv3 --pmc passes of scripts/mlp_bench.py)'

    # float32 cross-encoder: the hand-written kernel with the largest share of a step is ffn_out_split_kernel
    # in its feed-forward form (K = intermediate size): LayerNorm(residual + GELU(h + b1) W2^T + b2) with its
    # split output.  Algorithmic work of a launch over `rows` tokens: 2 * rows * K * hidden FLOPs (the kernel
    # executes three fp16 MFMA products per algorithmic one: priced against the fp16 matrix peak, executed rate
    # beside it) and rows * (4 K + 4 hidden [residual: the hi and lo halves of a split row] + 6 hidden [y as fp16
    # hi | hi | lo]) bytes; 4 hidden more with RAG_CHAIN_RESIDUAL=0, where an fp32 row is written as well.
    roofline_split = None
    if len(lls_ms):
        hid, inter = ce_cfg.hidden_size, ce_cfg.intermediate_size
        sel = lls_k == inter
        if sel.any():
            f_rows, f_s = float(lls_rows[sel].sum()), float(lls_ms[sel].sum()) / 1e3
            f_flops = 2.0 * f_rows * inter * hid
            y_row = 0.0 if os.environ.get('RAG_CHAIN_RESIDUAL', '1') != '0' else 4.0
            f_bytes = f_rows * (4.0 * inter + (10.0 + y_row) * hid)
            roofline_split = {'kernel': 'ffn_out_split_kernel<0> (second feed-forward product + bias + GELU + residual + LayerNorm + split, '
                                     'fp32-equivalent on fp16 MFMA)',
                           'bound': 'mfma', 'achieved': round(f_flops / f_s / 1e12, 1), 'peak': MFMA_BF16_PEAK_TFLOPS,
                           'unit': 'TFLOP/s', 'frac': round(f_flops / f_s / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), 'traffic': None,
                           'executed_tflops': round(3 * f_flops / f_s / 1e12, 1),
                           'executed_frac': round(3 * f_flops / f_s / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                           'vs_fp32_matrix_peak': round(f_flops / f_s / 1e12 / MFMA_F32_PEAK_TFLOPS, 3),
                           'launches_per_step': int(sel.sum()) // max(args.steps, 1),
                           'avg_launch_ms': round(f_s * 1e3 / int(sel.sum()), 3),
                           'avg_launch_rows': int(f_rows / int(sel.sum())),
                           'ms_per_step': round(f_s * 1e3 / max(args.steps, 1), 2),
                           'algorithmic_gbs': round(f_bytes / f_s / 1e9, 1),
                           'hbm_frac': round(f_bytes / f_s / 1e9 / HBM_PEAK_GBS, 4)}
            if len(lls_ms) >= 1024:      # the hooks' event ring was full: the figures cover the run's first launches only
                roofline_split['profile_ring_full'] = True
            if (~sel).any():      # the same kernel as attention-output projection (K = hidden, operand already split)
                o_rows, o_s = float(lls_rows[~sel].sum()), float(lls_ms[~sel].sum()) / 1e3
                roofline_split['attention_output_form'] = {
                    'ms_per_step': round(o_s * 1e3 / max(args.steps, 1), 2),
                    'algorithmic_gbs': round(o_rows * (14.0 + y_row) * hid / o_s / 1e9, 1),
                    'hbm_frac': round(o_rows * (14.0 + y_row) * hid / o_s / 1e9 / HBM_PEAK_GBS, 4),
                    'executed_tflops': round(6.0 * o_rows * hid * hid / o_s / 1e12, 1)}
            ff_pmc = os.path.join(ROOT, 'profiles', 'pmc_linear_ln_split.json')
            if os.path.exists(ff_pmc):
                with open(ff_pmc) as f:
                    per_row = float(json.load(f)['traffic_bytes_per_row'])
                roofline_split['traffic'] = round(per_row * f_rows / int(sel.sum()) / 1e9, 2)
                roofline_split['traffic_unit'] = ('GB per launch (bytes per row from rocprofv3 --pmc passes of '
                                               'scripts/split_out_bench.py, times the rows of an average launch)')

    result = {
        'metric': 'queries/sec end-to-end (embed+kNN+BM25+RRF+rerank) @ Recall@5, 1M x 768 corpus',
        'value': round(args.batch * args.steps / dt, 2), 'unit': 'queries/sec',
        'n_gpus': dist.get_world_size() if dist is not None else 1,
        'rccl_ranks': dist.get_world_size() if (dist is not None and dist.get_backend() == 'nccl') else 0,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32 (kNN) / f64 (BM25, RRF) / %s / %s' % (
            {'fp32': ('f32 cross-encoder forward with GEMMs and attention as 3 split-fp16 MFMA products (x = hi + lo, 22 significant '
                      'bits), fp32 accumulation: fp32 round-off class, see precision_check'
                      + ('; every pair screened first by an f16 forward, order and scores from the f32-equivalent one (rerank_screen)'
                         if ce_screen is not None else '')),
             'bf16': 'bf16 cross-encoder forward (does NOT hold the fp32 order: see precision_check)',
             'fp32-torch': 'f32 cross-encoder forward (PyTorch ops)'}[args.ce_dtype],
            'f32 (embedder forward)' if args.embed_fp32_gemm or args.embed_torch_ops else
            'f32 embedder forward with GEMMs as 3 split-bf16 MFMA products, fp32 accumulation (cosine within 1e-5 of fp32)'),
        'data': 'synthetic (seeded random unit vectors, Zipf postings, random token ids, random-init weights)',
        'config': {'workload': '%s x %d corpus%s, hybrid vector+BM25, RRF, MiniLM-L6 cross-encoder rerank_top_n=%d, '
                               'top_k=%d' % (('%gM' % (n_total / 1e6)) if n_total >= 1_000_000 else '%d rows' % n_total, args.dim,
                                             ' row-sharded %d ways' % world if world > 1 else '', t, args.top_k),
                   'rows_per_gpu': args.rows, 'rows_total': n_total, 'dim': args.dim, 'batch': args.batch,
                   'batch_per_gpu': batch_per_gpu, 'global_batch': args.batch,
                   'per_list_k': 2 * t, 'fused': t, 'ce_pairs_per_query': t, 'vocab': args.vocab,
                   'embedder': args.embed_arch, 'cross_encoder': args.ce_arch,
                   'parallelism': 'rows sharded x%d, all-gather top-k, pairs split' % world if world > 1 else 'single GPU',
                   'batch_pipelining': bool(world == 1 and args.pipeline and args.steps > 1)},
        'stage_ms_per_step': stage_ms,
        'rerank_screen': None,
        # `roofline`: the hand-written kernel with the largest share of a step (a feed-forward kernel of the cross-encoder:
        # the float16 one of the screening pass, or the split-precision one of the exact pass); the others beside it
        'roofline': None,
        'roofline_dense_scan': roofline,
    }
    ff = [r for r in (roofline_ff, roofline_split) if r is not None]
    ff.sort(key=lambda r: -r['ms_per_step'])
    result['roofline'] = ff[0] if ff else roofline
    if len(ff) > 1:
        result['roofline_second_feed_forward_kernel'] = ff[1]
    if searcher.screen_encoder is not None and screen_after['pairs'] > screen_before['pairs']:
        n_p = screen_after['pairs'] - screen_before['pairs']
        result['rerank_screen'] = {
            'screen': 'float16 copy of the cross-encoder on every pair (one MFMA product per product); '
                      + ("pairs within the margin of their query's k-th best cheap logit are rescored by the "
                         'float32-equivalent forward; ' if args.screen_one_stage else
                         "the k cheap-best of every query are rescored by the float32-equivalent forward, then the pairs whose "
                         'cheap logit, less the query\'s measured offset, comes within the margin of the k-th best EXACT logit '
                         'found so far; ')
                      + 'order and rerank_score come from the rescored logits only',
            'stages': 1 if args.screen_one_stage else 2,
            'rescored_fraction': round((screen_after['rescored'] - screen_before['rescored']) / n_p, 4),
            'pairs_per_step': n_p // max(args.steps, 1), 'safety': args.screen_safety,
            'within_query_error_max': searcher.screen_stats['err_max'],
            'margin': (2.0 if args.screen_one_stage else 1.0) * args.screen_safety * float(searcher._screen_err or 0.0),
            'margin_widened_in_timed_steps': screen_after['widened'] - screen_before['widened']}
    off_h = tok_h = None

    def search_with(srch, q):
        o = srch.search(*as_args(q))
        torch.cuda.synchronize()
        return o

    def time_steps(srch, batch_list, n_warm=1):
        """queries/sec and ms per step of `srch` over batch_list (after n_warm untimed batches of it)."""
        for q in batch_list[:n_warm]:
            srch.search(*as_args(q))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for q in batch_list:
            srch.search(*as_args(q))
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        nq = sum(int(q['ids'].shape[0]) for q in batch_list)
        return round(nq / el, 2), round(el / len(batch_list) * 1e3, 3)

    if world == 1 and not args.no_extras:
        # (1) the TIMED searcher against a plain PyTorch float32 cross-encoder (what CrossEncoder.predict computes,
        # reference src/reranker/rerank.py:79-84) on every query of batch 0: final top-k ids, in order.  A query is
        # excused only if two of the float32 logits that decide its top-k lie closer than eps (float32 round-off of
        # two different summation orders cannot be told apart there).
        eps = 2e-5
        o_timed = search_with(searcher, queries[0])
        exact_mask = searcher.last_rescored               # None: every logit is an exact one
        ce_ref = load_ce('fp32-torch')
        o_ref = search_with(make_searcher(ce_ref, t), queries[0])
        k = args.top_k
        ref_sorted = torch.sort(o_ref['logits'].double(), dim=1, descending=True).values
        gap = (ref_sorted[:, :k] - ref_sorted[:, 1:k + 1]).min(dim=1).values
        same = (o_timed['rows'] == o_ref['rows']).all(dim=1)
        fused_same = bool((o_timed['fused_row'] == o_ref['fused_row']).all())
        diff = (o_timed['logits'].double() - o_ref['logits'].double()).abs()
        err = float(diff.max() if exact_mask is None else diff[exact_mask].max())
        score_err = float((o_timed['rerank_score'].double() - o_ref['rerank_score'].double())[same].abs().max())
        result['precision_check'] = {
            'reference': 'plain PyTorch float32 cross-encoder forward on the same fused candidates',
            'queries': int(same.numel()), 'top_k_ids_identical': int(same.sum()),
            'differing_with_fp32_gap_below_eps': int((~same & (gap < eps)).sum()),
            'differing_unexcused': int((~same & (gap >= eps)).sum()), 'eps': eps,
            'fused_candidates_identical': fused_same, 'max_abs_logit_error': err,
            'max_abs_rerank_score_error': score_err,
            'logits_compared': int(diff.numel() if exact_mask is None else exact_mask.sum()),
            'rescored_fraction_of_this_batch': None if exact_mask is None else round(float(exact_mask.float().mean()), 4),
            'median_deciding_gap_fp32': float(gap.median())}
        del ce_ref

        # (2) secondary measurements, each a few timed steps of the same batches
        nx = max(1, min(args.extra_steps, len(queries)))
        extra = {}
        s30 = make_searcher(ce, 30, screen=ce_screen)
        v, ms = time_steps(s30, queries[:nx])
        extra['rerank_top_n_30'] = {'value': v, 'ms_per_step': ms, 'unit': 'queries/sec',
                                    'note': 'same path with --rerank_top_n 30 (60 entries per list, 30 pairs per query)'}
        if ce_screen is not None:
            st = s30.screen_stats
            # (the first batch of a fresh searcher rescores every pair: the fraction of the batches after it)
            extra['rerank_top_n_30']['rescored_fraction_all_batches'] = round(st['rescored'] / max(st['pairs'], 1), 4)
            v, ms = time_steps(make_searcher(ce, t), queries[:nx])
            extra['without_screen'] = {'value': v, 'ms_per_step': ms, 'unit': 'queries/sec',
                                       'note': 'every pair through the float32-equivalent forward (--ce-screen none)'}
        del s30
        # the pipeline every timing the reference publishes was taken on (evaluation/comprehensive_evaluation.py:233-241:
        # hybrid RRF, reranker_kind='simple', rerank_top_n=50, top_k=10; 1.37-1.85 s per query on 91,853 chunks there):
        # the simple reranker's cosines come from the resident rows (rag_dense_rescore_dev)
        s_simple = BatchSearcher(dense, bm25, canon=canon, bi_encoder=bi, cross_encoder=None, simple_rerank=True,
                                 top_k=10, rerank_top_n=50)
        v, ms = time_steps(s_simple, queries[:nx])
        extra['simple_reranker_top_n_50_top_k_10'] = {
            'value': v, 'ms_per_step': ms, 'unit': 'queries/sec',
            'note': "hybrid RRF + the reference's 'simple' (bi-encoder cosine) reranker, rerank_top_n=50, top_k=10: the "
                    "pipeline of the reference's own published per-query times (1.37-1.85 s per query on its 91,853-chunk "
                    "corpus, hardware unstated); here on the 1M-row shard, pool cosines from the resident rows"}
        del s_simple
        other = 'bf16' if args.ce_dtype != 'bf16' else 'fp32'
        ce_other = load_ce(other)
        s_other = make_searcher(ce_other, t)
        v, ms = time_steps(s_other, queries[:nx])
        o_other = search_with(s_other, queries[0])
        same_o = (o_other['rows'] == o_ref['rows']).all(dim=1)
        extra['cross_encoder_' + other] = {
            'value': v, 'ms_per_step': ms, 'unit': 'queries/sec',
            'top_k_ids_identical_to_fp32': '%d/%d' % (int(same_o.sum()), int(same_o.numel())),
            'max_abs_logit_error': float((o_other['logits'].double() - o_ref['logits'].double()).abs().max())}
        del ce_other, s_other
        if ce_screen is not None:              # the screening forward on its own: its speed and how often it alone would do
            s_h = make_searcher(ce_screen, t)
            v, ms = time_steps(s_h, queries[:nx])
            o_h = search_with(s_h, queries[0])
            same_h = (o_h['rows'] == o_ref['rows']).all(dim=1)
            extra['cross_encoder_fp16_alone'] = {
                'value': v, 'ms_per_step': ms, 'unit': 'queries/sec',
                'top_k_ids_identical_to_fp32': '%d/%d' % (int(same_h.sum()), int(same_h.numel())),
                'max_abs_logit_error': float((o_h['logits'].double() - o_ref['logits'].double()).abs().max())}
            del s_h
        # ---- the two configurations in which the retrieval kernels ARE the step --------------------------------------
        def timed_ms(fn, reps):
            """average milliseconds of fn() over reps calls, by HIP events on the current stream (after one untimed call)"""
            fn()
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            for _ in range(reps):
                fn()
            b_.record()
            b_.synchronize()
            return a.elapsed_time(b_) / reps

        def scan_roofline(index, n_rows, batch, ms):
            fl = 2.0 * n_rows * args.dim * batch
            byt = 4.0 * n_rows * args.dim
            r = {'kernel': 'dense scan (rag_dense_search_dev: pre-pass + main pass + list merge%s)' % (
                     ', split-bf16 products + fp32 rescoring' if getattr(index, 'split', False) else ''),
                 'avg_search_ms': round(ms, 4), 'unit': 'TFLOP/s' if batch >= 40 else 'GB/s'}
            if batch >= 40:              # matrix-core bound (arithmetic intensity batch / 2 FLOP per byte)
                split = bool(getattr(index, 'split', False))
                peak = MFMA_BF16_PEAK_TFLOPS if split else MFMA_F32_PEAK_TFLOPS
                ach = fl / (ms * 1e-3) / 1e12
                r.update(bound='mfma', achieved=round(ach, 1), peak=peak, frac=round(ach / peak, 4), traffic=None,
                         vs_fp32_matrix_peak=round(ach / MFMA_F32_PEAK_TFLOPS, 3))
                if split:
                    r.update(executed_tflops=round(3 * ach, 1), executed_frac=round(3 * ach / peak, 4))
            else:                        # one pass over the rows: HBM bound
                ach = byt / (ms * 1e-3) / 1e9
                r.update(bound='hbm', achieved=round(ach, 1), peak=HBM_PEAK_GBS, frac=round(ach / HBM_PEAK_GBS, 4), traffic=None,
                         note='algorithmic bytes = 4 x rows x dim per search' + (
                             '; %d MB of rows fit the 256 MB memory-side cache, so repeated searches need not reach HBM'
                             % (byt / 1e6) if byt < 256e6 * 1.5 else ''))
            return r

        # BASELINE configs[1]: 100k x 768, embed + brute-force cosine kNN top_k=10 (VectorIndexer.search,
        # reference src/indexing/vector_indexer.py:165-230), no BM25, no fusion, no reranker
        n2 = min(100_000, args.rows)
        dense2 = rag_hip.DenseIndex(rows[:n2], metric='cosine', borrow=True)
        dense2.reserve(args.batch, 10)
        s2 = BatchSearcher(dense2, None, bi_encoder=bi, top_k=10, rerank_top_n=10)
        q1 = make_queries(argparse.Namespace(**{**vars(args), 'batch': 1}), 6, dev, seed=411)

        def knn(q):
            return s2.dense_search(s2.embed(q['ids'], q['mask']), 10)
        qv_b = s2.embed(queries[0]['ids'], queries[0]['mask'])
        qv_1 = s2.embed(q1[0]['ids'], q1[0]['mask'])
        ms_e2e = timed_ms(lambda: knn(queries[0]), 20)
        ms_scan = timed_ms(lambda: s2.dense_search(qv_b, 10), 50)
        for q in q1:
            knn(q)                               # (single queries replay the embedder's HIP graph: shapes seen once first)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            for q in q1:
                knn(q)
        torch.cuda.synchronize()
        ms_one = (time.perf_counter() - t1) / (5 * len(q1)) * 1e3
        ms_scan1 = timed_ms(lambda: s2.dense_search(qv_1, 10), 200)
        extra['config_2_knn_100k'] = {
            'workload': '%d x %d corpus, %s embed + exact cosine kNN top_k=10' % (n2, args.dim, args.embed_arch),
            'value': round(args.batch / ms_e2e * 1e3, 1), 'unit': 'queries/sec', 'batch': args.batch,
            'ms_per_step': round(ms_e2e, 3), 'embed_ms': round(ms_e2e - ms_scan, 3),
            'single_query_latency_ms': round(ms_one, 3),
            'roofline': scan_roofline(dense2, n2, args.batch, ms_scan),
            'roofline_single_query': scan_roofline(dense2, n2, 1, ms_scan1)}
        del s2
        dense2.close()
        # BASELINE configs[2]: 1M x 768, hybrid vector + BM25 with RRF, top_k=5, no reranker
        # (HybridIndexer.search, reference src/indexing/hybrid_indexer.py:60-110: 2 x top_k entries per list)
        s3 = BatchSearcher(dense, bm25, canon=canon, bi_encoder=bi, top_k=args.top_k, rerank_top_n=args.top_k)
        v, ms = time_steps(s3, queries[:nx])
        k3 = 2 * args.top_k
        q0 = queries[0]
        qv3 = s3.embed(q0['ids'], q0['mask'])
        ms_d = timed_ms(lambda: s3.dense_search(qv3, k3), 20)
        ms_b = timed_ms(lambda: s3.bm25_search(q0['qterms'], q0['qoff'], q0['n_terms'], k3), 20)
        vec3, bm3 = s3.dense_search(qv3, k3), s3.bm25_search(q0['qterms'], q0['qoff'], q0['n_terms'], k3)
        ms_f = timed_ms(lambda: s3.fuse(vec3, bm3, args.top_k), 50)
        tq = q0['host']['terms']
        tq = tq[tq >= 0].astype(np.int64)                       # (out-of-vocabulary tokens have no postings)
        ip = np.asarray(host['arr'].indptr)
        postings = float((ip[tq + 1] - ip[tq]).sum())
        one3 = make_queries(argparse.Namespace(**{**vars(args), 'batch': 1}), 6, dev, seed=431)
        _, ms_one3 = time_steps(s3, one3, n_warm=len(one3))
        bm_pmc = os.path.join(ROOT, 'profiles', 'r03_pmc_bm25_sq.json')
        if not os.path.exists(bm_pmc):
            bm_pmc = os.path.join(ROOT, 'profiles', 'r02_pmc_bm25_sq.json')
        issue = None
        if os.path.exists(bm_pmc):
            with open(bm_pmc) as f:
                pm = json.load(f)
            ps = pm.get('per_search', {})
            issue = {'source': os.path.relpath(bm_pmc, ROOT), 'workload': pm.get('workload'),
                     'wave_time_shares': pm.get('wave_time_shares'),
                     'lds_bank_conflict_share_of_lds_active_cycles': (
                         round(ps['SQ_LDS_BANK_CONFLICT'] / ps['SQ_ACTIVE_INST_LDS'], 3)
                         if ps.get('SQ_ACTIVE_INST_LDS') else None)}
        extra['config_3_hybrid_rrf_1m'] = {
            'workload': '%d x %d corpus, hybrid vector + BM25, RRF, top_k=%d, no reranker' % (args.rows, args.dim, args.top_k),
            'value': v, 'unit': 'queries/sec', 'batch': args.batch, 'ms_per_step': ms,
            'single_query_latency_ms': ms_one3,
            'stage_ms': {'dense': round(ms_d, 3), 'bm25': round(ms_b, 3), 'fuse': round(ms_f, 3)},
            'roofline_dense_scan': scan_roofline(dense, args.rows, args.batch, ms_d),
            'bm25': {'kernel': 'bm25_tile_kernel (+ offsets, sample pass, list merges)', 'us_per_query': round(ms_b * 1e3 / args.batch, 3),
                     'postings_per_query': round(postings / args.batch, 1),
                     'algorithmic_gbs': round(12.0 * postings / (ms_b * 1e-3) / 1e9, 1),
                     'bound': 'instruction issue (the postings are L2 hits in tile-major order: the algorithmic rate is no HBM rate)',
                     'sq_counters': issue},
            'fuse_us_per_query': round(ms_f * 1e3 / args.batch, 3)}
        del s3
        sweep = []
        for bsz in (1, 32, 256):
            sub = argparse.Namespace(**{**vars(args), 'batch': bsz})
            qs = make_queries(sub, nx + 1, dev, seed=300 + bsz)
            # (small batches replay their encoder forwards from HIP graphs, captured per shape bucket the first time
            # a shape is seen: every batch of the list runs once before the list is timed, a warmed server's latency)
            v, ms = time_steps(searcher, qs[1:], n_warm=len(qs) - 1 if bsz <= 32 else 1)
            sweep.append({'batch': bsz, 'latency_ms': ms, 'value': v})
        extra['batch_sweep'] = sweep
        # the same step when the boundary hands over HOST buffers: queries copied in from pinned memory and the
        # result lists copied back inside the timed region
        pin = lambda x: x.cpu().pin_memory()
        hq = [{k2: pin(q[k2]) for k2 in ('ids', 'mask', 'qterms', 'qoff', 'ce_tok', 'ce_off')} | {'n_terms': q['n_terms']}
              for q in queries[:nx]]
        sink = None
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for q in hq:
            dq = {k2: (v2.to(dev, non_blocking=True) if hasattr(v2, 'to') else v2) for k2, v2 in q.items()}
            o = searcher.search(*as_args(dq))
            sink = [o[k2].to('cpu', non_blocking=True) for k2 in ('rows', 'scores', 'rerank_score', 'count')]
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        extra['host_buffers'] = {'value': round(args.batch * len(hq) / el, 2), 'ms_per_step': round(el / len(hq) * 1e3, 3),
                                 'note': 'H2D of the query tensors and D2H of rows / scores inside the timed region'}
        # index build (not part of a query step): the BM25 side of scripts/build_index.py on this shard's documents
        from indexing.bm25_indexer import build_arrays_device
        t1 = time.perf_counter()
        arr2 = build_arrays_device(host['term'], host['doc'], host['ln'], args.vocab, 1.5, 0.75, device=dev)
        torch.cuda.synchronize()
        t_csr = time.perf_counter() - t1
        words = ['w%05d' % i for i in range(20000)]
        rng_t = np.random.default_rng(3)
        n_txt = 50_000
        picks = np.minimum(rng_t.zipf(1.2, size=n_txt * 45) - 1, len(words) - 1)
        texts = [' '.join(words[j] for j in picks[i * 45:(i + 1) * 45]).encode('ascii') for i in range(n_txt)]
        t1 = time.perf_counter()
        ids_t, len_t, voc_t = rag_hip.bm25_tokenize(texts, np.zeros(n_txt, np.uint8))
        t_tok = time.perf_counter() - t1
        # the embedding side of the build: chunk texts of this shard through the bi-encoder (unpadded token streams,
        # the path MedicalEmbedder.encode takes on the GPU), as VectorIndexer.index_chunks does for every chunk
        n_emb = 16384
        e_off = off[:n_emb + 1].to(torch.int64)
        e_len = (e_off[1:] - e_off[:-1]).to(torch.int32)
        e_ids = tok[:int(e_off[-1])].contiguous()
        e_cu = torch.zeros((n_emb + 1,), dtype=torch.int32, device=dev)
        e_cu[1:] = torch.cumsum(e_len, 0)
        e_pos = (torch.arange(e_ids.shape[0], device=dev, dtype=torch.int32)
                 - torch.repeat_interleave(e_cu[:-1], e_len.long()))
        e_types = torch.zeros_like(e_ids)
        longest = int(e_len.max())
        with torch.no_grad():
            bi.forward_packed(e_ids, e_types, e_pos, e_cu, longest, normalize=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            bi.forward_packed(e_ids, e_types, e_pos, e_cu, longest, normalize=True)
            torch.cuda.synchronize()
        t_emb = time.perf_counter() - t1
        extra['index_build'] = {
            'embed_chunks_per_s': round(n_emb / t_emb, 1), 'embed_tokens_per_s': round(int(e_ids.shape[0]) / t_emb, 1),
            'embed_sample': '%d chunks, %d tokens, %s forward (fp32, split-bf16 GEMMs) + mean pooling + L2 norm in one '
                            'unpadded batch' % (n_emb, int(e_ids.shape[0]), args.embed_arch),
            'bm25_postings_build_s': round(t_csr, 3), 'documents': int(host['ln'].shape[0]), 'tokens': int(host['term'].shape[0]),
            'postings': int(arr2.rows.shape[0]),
            'note': 'term-id stream -> term-major CSR + fp64 impacts on the device (rag_bm25_build_dev, rag_bm25_impacts_dev), '
                    'host copies included',
            'native_tokenizer_mtokens_per_s': round(ids_t.shape[0] / t_tok / 1e6, 2),
            'native_tokenizer_sample': '%d ASCII documents, %d tokens, one host core (rag_bm25_tokenize_*)' % (n_txt, ids_t.shape[0])}
        del arr2
        result['secondary'] = extra

    if args.cpu_queries > 0 and world == 1:
        bi_cpu, _, _ = bert.load_model('random:' + args.embed_arch, 'bi', 'cpu')
        ce_cpu, _, _ = bert.load_model('random:' + args.ce_arch, 'cross', 'cpu')
        # the oracle's ids against those of the TIMED searcher (same object, same kernels) on batch 0
        o = search_with(searcher, queries[0])
        off_h = off.cpu().numpy()
        tok_h = tok.cpu().numpy()
        gpu_out = {'fused_row': o['fused_row'].cpu().numpy(), 'rows': o['rows'].cpu().numpy(),
                   'chunk_tokens': lambda rs: [tok_h[off_h[r]:off_h[r + 1]].tolist() for r in rs]}
        result['cpu_baseline'] = cpu_baseline(args, (bi_cpu, ce_cpu), rows, host, queries[0], gpu_out)
    rag_hip.check_faults()          # every device-side consistency check held during the run
    result['config']['gemm_solution_table'] = bool(tuned)
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
