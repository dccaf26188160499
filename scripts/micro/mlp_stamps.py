#!/usr/bin/env python3
"""In-kernel cycle stamps of mlp_fused_kernel (a library built with -DML_STAMPS=1, given by RAG_HIP_LIB): cycles of one
workgroup's wave 1 per chunk iteration (set-up, the 24 k-steps, the closing wait for the weight loads, the barrier),
the loop, the prologue and the epilogue.

    RAG_HIP_LIB=build/librag_stamps.so python scripts/micro/mlp_stamps.py [--dtype fp16]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'domain-rag-system_amd'))

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='fp16', choices=['bf16', 'fp16'])
    ap.add_argument('--tokens', type=int, default=1 << 20)
    a = ap.parse_args()
    import rag_hip
    dev = torch.device('cuda:0')
    lib = rag_hip.init(0)
    g = torch.Generator(device=dev).manual_seed(1)
    dt = torch.bfloat16 if a.dtype == 'bf16' else torch.float16
    x = torch.randn((a.tokens, 384), generator=g, device=dev).to(dt)
    l1 = torch.nn.Linear(384, 1536, device=dev, dtype=dt)
    l2 = torch.nn.Linear(1536, 384, device=dev, dtype=dt)
    gamma, beta = torch.ones((384,), device=dev, dtype=dt), torch.zeros((384,), device=dev, dtype=dt)
    with torch.no_grad():
        for _ in range(3):
            rag_hip.enc_mlp_dev(x, l1.weight, l1.bias, l2.weight, l2.bias, gamma, beta)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    lib.rag_enc_mlp_debug.argtypes = [C.c_void_p]
    assert lib.rag_enc_mlp_debug(out) == 0
    v = list(out)
    n = max(v[4], 1)
    print({'per_chunk': {'setup': v[0] / n, 'k_steps': v[1] / n, 'wait_for_loads': v[2] / n, 'barrier': v[3] / n,
                         'sum': (v[0] + v[1] + v[2] + v[3]) / n}, 'chunks': v[4], 'loop': v[5], 'prologue': v[6],
           'epilogue': v[7], 'tile': v[5] + v[6] + v[7]})


if __name__ == '__main__':
    main()
