#!/usr/bin/env python3
"""The fp16-in / fp32-out library GEMMs of the float32 cross-encoder under PyTorch's two BLAS back ends.

    python scripts/micro/mixed_gemm_backends.py --rows 655360
"""
import argparse
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=655360)
    ap.add_argument('--iters', type=int, default=5)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    m = args.rows
    g = torch.Generator(device=dev).manual_seed(0)

    def timed(fn):
        fn(); fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / args.iters
    for n in (1152, 1536):
        x = (torch.randn((m, 1152), generator=g, device=dev) * 0.5).half()
        w = (torch.randn((n, 1152), generator=g, device=dev) * 0.05).half()
        flops = 2.0 * m * 1152 * n
        for lib in ('hipblaslt', 'cublas', 'ck'):
            try:
                torch.backends.cuda.preferred_blas_library(lib)
                ms = timed(lambda: torch.mm(x, w.t(), out_dtype=torch.float32))
                ms16 = timed(lambda: torch.mm(x, w.t()))
                print(f'n={n} {lib:9s}: fp32 out {ms:.3f} ms = {flops / ms / 1e9:.0f} TFLOP/s; fp16 out {ms16:.3f} ms = {flops / ms16 / 1e9:.0f} TFLOP/s', flush=True)
            except Exception as e:  # noqa: BLE001
                print(f'n={n} {lib}: {type(e).__name__}: {str(e)[:200]}', flush=True)
        # the weight as [K][N] row-major (x @ w2) instead of the transposed view of [N][K]
        torch.backends.cuda.preferred_blas_library('hipblaslt')
        w2 = w.t().contiguous()
        ms = timed(lambda: torch.mm(x, w2, out_dtype=torch.float32))
        print(f'n={n} hipblaslt, weight stored [K][N]: fp32 out {ms:.3f} ms = {flops / ms / 1e9:.0f} TFLOP/s', flush=True)


if __name__ == '__main__':
    main()
