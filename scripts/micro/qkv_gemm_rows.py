#!/usr/bin/env python3
"""The fp16-in / fp32-out q|k|v product (N = K = 1152) at different row counts: the library's heuristic changes tiles."""
import torch

dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
N = int(__import__("sys").argv[1]) if len(__import__("sys").argv) > 1 else 1152
w = (torch.randn((N, 1152), generator=g, device=dev) * 0.05).half()
total = 786432
x_all = (torch.randn((total, 1152), generator=g, device=dev) * 0.5).half()
out = torch.empty((total, N), dtype=torch.float32, device=dev)
for rows in (786432, 393216, 262144, 131072, 65536, 49152, 32768, 24576, 16384):
    def run():
        for r0 in range(0, total, rows):
            torch.mm(x_all[r0:r0 + rows], w.t(), out_dtype=torch.float32, out=out[r0:r0 + rows])
    run(); run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f'rows per call {rows:7d}: {ms:.3f} ms for {total} rows = {2.0 * total * 1152 * N / ms / 1e9:.0f} TFLOP/s', flush=True)
