#!/usr/bin/env python3
"""rag_enc_attention_split_dev alone on a packed batch with the bench's pair lengths (longest first).

    python scripts/micro/att_split_bench.py [--tokens 1000000]      (RAG_HIP_LIB=... for A/B builds)
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'domain-rag-system_amd'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tokens', type=int, default=1_000_000)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--mean', type=float, default=45.0)
    ap.add_argument('--clip', type=int, default=0, help='cut every sequence to this many tokens')
    ap.add_argument('--hi-lo', action='store_true', help='[hi | lo] output rows (rag_enc_attention_split_hl_dev)')
    ap.add_argument('--lds-for', type=int, default=0, help='size the LDS image for this sequence length (>= the longest)')
    args = ap.parse_args()
    import rag_hip
    rag_hip.init(0)
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(0)
    lens = []
    while sum(lens) < args.tokens:
        chunk = int(np.clip(np.rint(rng.lognormal(np.log(args.mean), 0.6)), 2, 128)) + 2
        lens.append(chunk + int(rng.integers(8, 25)) + 3)
    lens = np.sort(np.asarray(lens))[::-1].copy()
    if args.clip:
        lens = np.minimum(lens, args.clip)
    t = int(lens.sum())
    cu = torch.zeros(len(lens) + 1, dtype=torch.int32, device=dev)
    cu[1:] = torch.cumsum(torch.from_numpy(lens).to(dev), 0)
    qkv = torch.randn((t, 3 * 384), device=dev)
    bias = torch.randn((3 * 384,), device=dev) * 0.1
    longest = max(int(lens.max()), args.lds_for)
    for first in (False,):
        fn = lambda: rag_hip.enc_attention_split_dev(qkv, bias, cu, longest, 12, first_only=first, half=torch.float16, hi_lo=args.hi_lo)
        fn(); fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / args.iters
        gb = t * (4608 + 2304) / 1e9
        print(f'attention_split: {len(lens)} sequences, {t} tokens, longest {longest}: {ms:.3f} ms, {gb / ms * 1e3:.0f} GB/s algorithmic', flush=True)


if __name__ == '__main__':
    main()
