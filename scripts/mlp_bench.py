#!/usr/bin/env python3
"""The one-kernel feed-forward block alone (for rocprofv3 --pmc passes and A/B runs).

    python scripts/mlp_bench.py --tokens 1048576 --iters 5 [--dtype fp16] [--only-mlp]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'domain-rag-system_amd'))

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tokens', type=int, default=1 << 20)
    ap.add_argument('--inter', type=int, default=1536)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp16'], help='element type of the kernel instance')
    ap.add_argument('--only-mlp', action='store_true', help='time the feed-forward kernel only (for counter passes)')
    args = ap.parse_args()
    import rag_hip
    dev = torch.device('cuda:0')
    rag_hip.init(0)
    g = torch.Generator(device=dev).manual_seed(1)
    bf = torch.bfloat16 if args.dtype == 'bf16' else torch.float16
    x = torch.randn((args.tokens, 384), generator=g, device=dev).to(bf)
    l1 = torch.nn.Linear(384, args.inter, device=dev, dtype=bf)
    l2 = torch.nn.Linear(args.inter, 384, device=dev, dtype=bf)
    gamma = torch.ones((384,), device=dev, dtype=bf)
    beta = torch.zeros((384,), device=dev, dtype=bf)
    with torch.no_grad():
        rag_hip.enc_mlp_dev(x, l1.weight, l1.bias, l2.weight, l2.bias, gamma, beta)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            rag_hip.enc_mlp_dev(x, l1.weight, l1.bias, l2.weight, l2.bias, gamma, beta)
        b.record()
        torch.cuda.synchronize()
    ms = a.elapsed_time(b) / args.iters
    fl = 4.0 * args.tokens * 384 * args.inter
    print(f'feed-forward block: tokens={args.tokens} inter={args.inter} {ms:.3f} ms, {fl / ms / 1e9:.0f} TFLOP/s, '
          f'{args.tokens * 384 * 2 * 2 / ms / 1e6:.0f} GB/s algorithmic', flush=True)
    if args.only_mlp:
        rag_hip.check_faults()
        return
    lo = torch.nn.Linear(384, 384, device=dev, dtype=bf)
    ctx = torch.randn((args.tokens, 384), generator=g, device=dev).to(bf)

    def layer_by_layer():
        x1 = rag_hip.enc_add_layernorm_dev(lo(ctx), x, gamma, beta, 1e-12)
        return rag_hip.enc_mlp_dev(x1, l1.weight, l1.bias, l2.weight, l2.bias, gamma, beta)

    def one_kernel():
        return rag_hip.enc_attn_out_mlp_dev(ctx, x, lo.weight, lo.bias, gamma, beta, l1.weight, l1.bias, l2.weight,
                                            l2.bias, gamma, beta)
    with torch.no_grad():
        for name, fn in (('projection GEMM + LayerNorm kernel + feed-forward kernel', layer_by_layer),
                         ('attention output + feed-forward in one kernel', one_kernel)):
            fn()
            torch.cuda.synchronize()
            a.record()
            for _ in range(args.iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            print(f'{name}: {a.elapsed_time(b) / args.iters:.3f} ms', flush=True)
    lq = torch.nn.Linear(384, 1152, device=dev, dtype=bf)
    with torch.no_grad():
        for name, fn in (('q|k|v projection, library GEMM', lambda: lq(x)),
                         ('q|k|v projection, rag_enc_linear384_dev', lambda: rag_hip.enc_linear384_dev(x, lq.weight, lq.bias))):
            fn()
            torch.cuda.synchronize()
            a.record()
            for _ in range(args.iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / args.iters
            print(f'{name}: {ms:.3f} ms ({2.0 * args.tokens * 384 * 1152 / ms / 1e9:.0f} TFLOP/s)', flush=True)
    rag_hip.check_faults()
    import ctypes
    L = rag_hip.lib()
    if hasattr(L, 'rag_enc_mlp_debug'):
        buf = (ctypes.c_ulonglong * 8)()
        L.rag_enc_mlp_debug(buf)
        n = max(buf[4], 1)
        print('stamps per chunk (cycles): dma issue %.0f, steps %.0f, vmcnt wait %.0f, barrier %.0f; loop total per chunk %.0f; per tile: prologue %d, loop %d, epilogue %d' % (buf[0] / n, buf[1] / n, buf[2] / n, buf[3] / n, buf[5] / n, buf[6], buf[5], buf[7]))


if __name__ == '__main__':
    main()
