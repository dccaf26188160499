#!/usr/bin/env python3
"""Kernel-level timings of the hand-written stages at full size (no transformer forwards).

    python scripts/kernel_bench.py --rows 1000000 --batch 1024 --k 100 --iters 5 [--stage dense|bm25|all]

Used for rocprofv3 --pmc passes and for A/B runs of kernel variants.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'domain-rag-system_amd'))

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=1_000_000)
    ap.add_argument('--dim', type=int, default=768)
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--k', type=int, default=100)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--vocab', type=int, default=200_000)
    ap.add_argument('--stage', default='all', choices=['dense', 'bm25', 'build', 'all'])
    args = ap.parse_args()
    import rag_hip
    from tests import synth
    dev = torch.device('cuda:0')
    rag_hip.init(0)
    g = torch.Generator(device=dev).manual_seed(1)

    if args.stage in ('dense', 'all'):
        rows = torch.nn.functional.normalize(torch.randn((args.rows, args.dim), generator=g, device=dev), dim=1)
        q = torch.nn.functional.normalize(torch.randn((args.batch, args.dim), generator=g, device=dev), dim=1)
        idx = rag_hip.DenseIndex(rows, borrow=True)
        idx.search_dev(q, args.k)
        torch.cuda.synchronize()
        idx.profile(True)
        t0 = time.perf_counter()
        for _ in range(args.iters):
            idx.search_dev(q, args.k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.iters
        ms = idx.profile_read()
        flops = 2.0 * args.rows * args.dim * args.batch
        print(f'dense: rows={args.rows} batch={args.batch} k={args.k} wall={dt * 1e3:.3f} ms/search '
              f'scan={ms.mean():.3f} ms (min {ms.min():.3f}) -> {flops / ms.mean() / 1e9:.1f} TFLOP/s, '
              f'{4.0 * args.rows * args.dim / ms.mean() / 1e6:.1f} GB/s algorithmic, '
              f'{args.batch / dt:.0f} q/s', flush=True)

    if args.stage in ('bm25', 'all'):
        from indexing.bm25_indexer import build_arrays
        term, doc, ln = synth.sparse_corpus(args.rows, args.vocab, seed=2000)
        arr = build_arrays(term, doc, ln, args.vocab, 1.5, 0.75)
        bm = rag_hip.Bm25Index(arr.indptr, arr.rows, arr.impacts(), arr.idf, args.rows)
        rng = np.random.default_rng(99)
        cdf = np.cumsum(synth.zipf_probs(args.vocab))
        nt = rng.integers(3, 13, size=args.batch)
        terms = np.minimum(np.searchsorted(cdf, rng.random(int(nt.sum()))), args.vocab - 1).astype(np.int32)
        qoff = np.zeros(args.batch + 1, np.int32)
        qoff[1:] = np.cumsum(nt)
        df = np.diff(arr.indptr)
        postings = int(df[terms].sum())
        tq, to = torch.from_numpy(terms).to(dev), torch.from_numpy(qoff).to(dev)
        bm.search_dev(tq, to, int(nt.sum()), args.k)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            bm.search_dev(tq, to, int(nt.sum()), args.k)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / args.iters
        print(f'bm25: docs={args.rows} batch={args.batch} k={args.k} {ms:.3f} ms/search, postings/query='
              f'{postings / args.batch:.0f}, algorithmic {12.0 * postings / ms / 1e6:.1f} GB/s, '
              f'{args.batch / ms * 1e3:.0f} q/s', flush=True)

    if args.stage == 'build':
        # index build: host counting (numpy) against the device build of the same token stream
        from indexing.bm25_indexer import build_arrays
        term, doc, ln = synth.sparse_corpus(args.rows, args.vocab, seed=2000)
        t0 = time.perf_counter()
        ref = build_arrays(term, doc, ln, args.vocab, 1.5, 0.75)
        ref_imp = ref.impacts()
        host_s = time.perf_counter() - t0
        term_d = torch.from_numpy(term.astype(np.int32)).to(dev)
        off = np.zeros(args.rows + 1, np.int64)
        np.cumsum(ln, out=off[1:])
        off_d, ln_d = torch.from_numpy(off).to(dev), torch.from_numpy(ln).to(dev)
        rag_hip.bm25_build_dev(term_d, off_d, args.vocab)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            ip, rows, tf = rag_hip.bm25_build_dev(term_d, off_d, args.vocab)
            imp = rag_hip.bm25_impacts_dev(rows, tf, ln_d, ref.avgdl, 1.5, 0.75)
        torch.cuda.synchronize()
        dev_s = (time.perf_counter() - t0) / args.iters
        same = (np.array_equal(ip.cpu().numpy(), ref.indptr) and np.array_equal(rows.cpu().numpy().view(np.uint32), ref.rows)
                and np.array_equal(tf.cpu().numpy().view(np.uint32), ref.tf)
                and np.array_equal(imp.cpu().numpy().view(np.uint64), ref_imp.view(np.uint64)))
        n_tok, nnz = int(term.shape[0]), int(rows.shape[0])
        # algorithmic bytes: tokens read twice (4 B), the posting stream written and sorted once
        # (12 B out + 12 B in), rows / tf / impacts written (16 B)
        gb = (8.0 * n_tok + 40.0 * nnz) / 1e9
        print(f'build: docs={args.rows} tokens={n_tok} postings={nnz}: host {host_s:.2f} s, device {dev_s * 1e3:.1f} ms '
              f'({n_tok / dev_s / 1e9:.2f} G tokens/s, {gb / dev_s:.0f} GB/s algorithmic), arrays identical: {same}', flush=True)
    rag_hip.check_faults()


if __name__ == '__main__':
    main()
