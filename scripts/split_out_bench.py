#!/usr/bin/env python3
"""rag_enc_linear_ln_split_dev alone against the route it replaces (library GEMM on [hi|hi|lo] operands with the
bias / GELU / split pass in front and the bias / residual / LayerNorm / split pass behind), for rocprofv3 --pmc
passes and A/B runs.

    python scripts/split_out_bench.py --tokens 1048576 --k 1536 --iters 5
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
    ap.add_argument('--k', type=int, default=1536, help='input width: 1536 = second feed-forward product, 384 = output projection')
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--abl', default=None, help='comma list of RAG_FS_ABL masks to time (library built with -DFS_STAMPS=1)')
    ap.add_argument('--forms', action='store_true', help='time the residual / output forms of both operand sources and stop')
    ap.add_argument('--only', default=None, choices=['fused', 'fused-split', 'fused-chain'],
                    help='run only the fused kernel, fp32 input or pre-split input (profiler passes)')
    args = ap.parse_args()
    import rag_hip
    dev = torch.device('cuda:0')
    rag_hip.init(0)
    g = torch.Generator(device=dev).manual_seed(1)
    m, k = args.tokens, args.k
    h = torch.randn((m, k), generator=g, device=dev)
    b1 = torch.randn((k,), generator=g, device=dev) * 0.1
    w = torch.randn((384, k), generator=g, device=dev) * 0.04
    b2 = torch.zeros((384,), device=dev)
    gamma, beta = torch.ones((384,), device=dev), torch.zeros((384,), device=dev)
    res = torch.randn((m, 384), generator=g, device=dev)
    wh, wl = rag_hip.split_weight_halves(w)
    w3 = rag_hip.split_weight_bf16x3(w, torch.float16)
    f16 = torch.float16

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / args.iters
    flops = 2.0 * m * k * 384

    def stamps():
        import ctypes
        buf = (ctypes.c_ulonglong * 8)()
        rag_hip.load_library().rag_enc_linear_ln_split_debug(buf)
        return 'prologue %d, loop %d (%d per chunk), epilogue %d' % (buf[0], buf[1], buf[1] // max(buf[3], 1), buf[2])
    def check():
        """max error of both operand forms against float64 on the first 2048 + last 100 tokens"""
        sel = torch.cat([torch.arange(0, min(m, 2048), device=dev), torch.arange(max(m - 100, 0), m, device=dev)])
        y0, ys0 = rag_hip.enc_linear_ln_split_dev(h, b1, wh, wl, b2, res, gamma, beta, 1e-12)
        a3_ = rag_hip.enc_split_bf16x3_dev(h, torch.float16)
        y1, _ = rag_hip.enc_linear_ln_split_dev(a3_, None, wh, wl, b2, res, gamma, beta, 1e-12)
        hd, wd = h[sel].double(), w.double()
        ref = lambda a: torch.nn.functional.layer_norm(res[sel].double() + a @ wd.t() + b2.double(), (384,), gamma.double(), beta.double(), 1e-12)
        e0 = float((y0[sel].double() - ref(torch.nn.functional.gelu(hd + b1.double()))).abs().max())
        e1 = float((y1[sel].double() - ref(hd)).abs().max())
        n = 384
        es = float(((ys0[sel, :n].double() + ys0[sel, 2 * n:].double()) - y0[sel].double()).abs().max())
        return f'max error vs float64: fp32-in {e0:.2e}, split-in {e1:.2e}; split output hi + lo - y: {es:.2e}'
    if args.abl:
        print(check(), flush=True)
        a3 = rag_hip.enc_split_bf16x3_dev(h, torch.float16)
        with torch.no_grad():
            for mask in args.abl.split(','):
                os.environ['RAG_FS_ABL'] = mask
                ms0 = timed(lambda: rag_hip.enc_linear_ln_split_dev(h, b1, wh, wl, b2, res, gamma, beta, 1e-12))
                s0 = stamps()
                ms1 = timed(lambda: rag_hip.enc_linear_ln_split_dev(a3, None, wh, wl, b2, res, gamma, beta, 1e-12))
                print(f'abl {mask:>3}: fp32-in {ms0:.3f} ms [{s0}]   split-in {ms1:.3f} ms [{stamps()}]', flush=True)
        return
    if args.forms:
        a3 = rag_hip.enc_split_bf16x3_dev(h, f16)
        a2 = torch.cat([a3[:, :k], a3[:, 2 * k:]], dim=1).contiguous()
        r3 = rag_hip.enc_split_bf16x3_dev(res, f16)
        fn = rag_hip.enc_linear_ln_split_dev
        with torch.no_grad():
            for name, src, bias in (('fp32 in, GELU', h, b1), ('[hi|hi|lo] in', a3, None), ('[hi|lo] in', a2, None)):
                t0 = timed(lambda: fn(src, bias, wh, wl, b2, res, gamma, beta, 1e-12))
                t1 = timed(lambda: fn(src, bias, wh, wl, b2, r3, gamma, beta, 1e-12))
                t2 = timed(lambda: fn(src, bias, wh, wl, b2, r3, gamma, beta, 1e-12, want_y=False))
                t3 = timed(lambda: fn(src, bias, wh, wl, b2, res, gamma, beta, 1e-12, want_split=False))
                t4 = timed(lambda: fn(src, bias, wh, wl, b2, res, gamma, beta, 1e-12, want_y=False))
                print(f'k={k} {name:14s}: fp32 residual, y + split {t0:.3f} ms | split residual, y + split {t1:.3f} | '
                      f'split residual, split only {t2:.3f} | fp32 residual, y only {t3:.3f} | fp32 residual, split only {t4:.3f}',
                      flush=True)
        return
    if args.only == 'fused-chain':      # the form a layer of the cross-encoder runs: fp32 in + GELU, residual = a split row, split row out
        r3 = rag_hip.enc_split_bf16x3_dev(res, f16)
        with torch.no_grad():
            ms = timed(lambda: rag_hip.enc_linear_ln_split_dev(h, b1, wh, wl, b2, r3, gamma, beta, 1e-12, want_y=False))
        print(f'fused (fp32 in, GELU; split residual, split out): tokens={m} k={k} {ms:.3f} ms  {3 * flops / ms / 1e9:.0f} TFLOP/s executed',
              flush=True)
        return
    if args.only == 'fused-split':
        a3 = rag_hip.enc_split_bf16x3_dev(h, torch.float16)
        with torch.no_grad():
            ms = timed(lambda: rag_hip.enc_linear_ln_split_dev(a3, None, wh, wl, b2, res, gamma, beta, 1e-12))
        print(f'fused (pre-split in): tokens={m} k={k} {ms:.3f} ms  {3 * flops / ms / 1e9:.0f} TFLOP/s executed', flush=True)
        return
    with torch.no_grad():
        ms = timed(lambda: rag_hip.enc_linear_ln_split_dev(h, b1, wh, wl, b2, res, gamma, beta, 1e-12))
        print(f'fused (fp32 in, GELU): tokens={m} k={k} {ms:.3f} ms  {flops / ms / 1e9:.0f} TFLOP/s algorithmic, '
              f'{3 * flops / ms / 1e9:.0f} executed', flush=True)
        if hasattr(rag_hip.load_library(), 'rag_enc_linear_ln_split_debug'):
            import ctypes
            buf = (ctypes.c_ulonglong * 8)()
            rag_hip.load_library().rag_enc_linear_ln_split_debug(buf)
            print('cycles of one workgroup: prologue %d, loop %d (%d chunks: %d per chunk), epilogue %d' % (
                buf[0], buf[1], buf[3], buf[1] // max(buf[3], 1), buf[2]), flush=True)
        if args.only:
            return
        a3 = rag_hip.enc_split_bf16x3_dev(h, f16)
        ms = timed(lambda: rag_hip.enc_linear_ln_split_dev(a3, None, wh, wl, b2, res, gamma, beta, 1e-12))
        print(f'fused (pre-split in):  {ms:.3f} ms  {3 * flops / ms / 1e9:.0f} TFLOP/s executed', flush=True)

        def separate():
            hs = rag_hip.enc_bias_act_split_dev(h, b1, True, f16)
            y = torch.mm(hs, w3.t(), out_dtype=torch.float32)
            return rag_hip.enc_bias_add_layernorm_split_dev(y, b2, res, gamma, beta, 1e-12, half=f16)
        ms = timed(separate)
        print(f'separate (GELU+split pass, library GEMM, LayerNorm+split pass): {ms:.3f} ms', flush=True)
        ms = timed(lambda: torch.mm(a3, w3.t(), out_dtype=torch.float32))
        print(f'library GEMM alone: {ms:.3f} ms  {3 * flops / ms / 1e9:.0f} TFLOP/s executed', flush=True)
    rag_hip.check_faults()


if __name__ == '__main__':
    main()
