#!/usr/bin/env python3
"""The 16-bit kernels between the feed-forward blocks of the screening forward, each alone, on a packed batch with the
bench's pair lengths (longest first, as the packed route orders them): rag_enc_attention_packed_dev and the two shapes
of rag_enc_linear384_dev (q|k|v projection, attention-output projection).

    python scripts/micro/att16_bench.py [--tokens 1000000] [--dtype fp16]      (RAG_HIP_LIB=... for A/B builds)
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'domain-rag-system_amd'))


def timed(fn, iters):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tokens', type=int, default=1_000_000)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--mean', type=float, default=45.0)
    ap.add_argument('--dtype', default='fp16', choices=['fp16', 'bf16'])
    ap.add_argument('--only', default='', help='attention | qkv | out | fused')
    ap.add_argument('--clip', type=int, default=0, help='cut every sequence to this many tokens')
    ap.add_argument('--uniform', type=int, default=0, help='every sequence this long (no query block straddles two sequences when it divides 32)')
    ap.add_argument('--pack', action='store_true', help='order the sequences for full tiles instead of longest first')
    args = ap.parse_args()
    import rag_hip
    rag_hip.init(0)
    dev = torch.device('cuda:0')
    dt = torch.float16 if args.dtype == 'fp16' else torch.bfloat16
    rng = np.random.default_rng(0)
    lens = []
    while sum(lens) < args.tokens:
        chunk = int(np.clip(np.rint(rng.lognormal(np.log(args.mean), 0.6)), 2, 128)) + 2
        lens.append(chunk + int(rng.integers(8, 25)) + 3)
    lens = np.sort(np.asarray(lens))[::-1].copy()
    if args.clip:
        lens = np.minimum(lens, args.clip)
    if args.uniform:
        lens = np.full(args.tokens // args.uniform, args.uniform)
    if args.pack:                                          # the order that fills the fused kernel's tiles
        order, packed_tiles = rag_hip.enc_qkv_attention_pack(lens)
        lens = lens[order]
    t = int(lens.sum())
    cu = torch.zeros(len(lens) + 1, dtype=torch.int32, device=dev)
    cu[1:] = torch.cumsum(torch.from_numpy(lens).to(dev), 0)
    qkv = torch.randn((t, 3 * 384), device=dev).to(dt)
    x = torch.randn((t, 384), device=dev).to(dt)
    wqkv = (torch.randn((1152, 384), device=dev) * 0.05).to(dt)
    wo = (torch.randn((384, 384), device=dev) * 0.05).to(dt)
    bqkv = torch.zeros(1152, device=dev, dtype=dt)
    bo = torch.zeros(384, device=dev, dtype=dt)
    longest = int(lens.max())
    cases = {'attention': (lambda: rag_hip.enc_attention_packed_dev(qkv, cu, longest, 12), t * (2304 + 768)),
             'qkv': (lambda: rag_hip.enc_linear384_dev(x, wqkv, bqkv), t * (768 + 2304)),
             'out': (lambda: rag_hip.enc_linear384_dev(x, wo, bo), t * (768 + 768))}
    tiles = torch.from_numpy(packed_tiles if args.pack else rag_hip.enc_qkv_attention_tiles(np.concatenate([[0], np.cumsum(lens)]))).to(dev)
    cases['fused'] = (lambda: rag_hip.enc_qkv_attention_dev(x, wqkv, bqkv, cu, tiles, 12), t * (768 + 768))
    if not args.only or args.only == 'fused':
        want = rag_hip.enc_attention_packed_dev(rag_hip.enc_linear384_dev(x, wqkv, bqkv), cu, longest, 12)
        got = rag_hip.enc_qkv_attention_dev(x, wqkv, bqkv, cu, tiles, 12)
        torch.cuda.synchronize()
        rag_hip.check_faults()
        d = (got.float() - want.float()).abs()
        print(f'fused vs projection + attention: {tiles.shape[0] - 1} tiles, max abs difference {float(d.max()):.3e}, '
              f'mean {float(d.mean()):.3e}, mean |value| {float(want.float().abs().mean()):.3e}', flush=True)
    for name, (fn, nbytes) in cases.items():
        if args.only and name != args.only:
            continue
        ms = timed(fn, args.iters)
        if name == 'fused' and hasattr(rag_hip.lib(), 'rag_enc_qkv_attention_debug'):   # a library built with -DQA_STAMPS=1
            import ctypes as C
            out = (C.c_ulonglong * 8)()
            rag_hip.lib().rag_enc_qkv_attention_debug.argtypes = [C.c_void_p]
            assert rag_hip.lib().rag_enc_qkv_attention_debug(out) == 0
            v = list(out)
            print({'per_tile': {'k_steps_36_images': v[0], 'image_tails': v[1], 'waits_and_barriers': v[2], 'attention_12_heads': v[3],
                                'prologue': v[4], 'workgroup': v[5]}, 'key_span': v[6], 'tokens': v[7]})
        if name in ('qkv', 'out') and hasattr(rag_hip.lib(), 'rag_enc_linear_debug'):      # a library built with -DLN_STAMPS=1
            import ctypes as C
            out = (C.c_ulonglong * 8)()
            rag_hip.lib().rag_enc_linear_debug.argtypes = [C.c_void_p]
            assert rag_hip.lib().rag_enc_linear_debug(out) == 0
            v = list(out)
            n = max(v[3], 1)
            print({'per_image': {'k_steps': v[0] / n, 'bias_and_pack': v[1] / n, 'wait_and_barrier': v[2] / n}, 'images': v[3],
                   'prologue': v[4], 'loop_and_tail': v[5], 'workgroup': v[6], 'workgroup_100MHz_ticks': v[7],
                   'clock_GHz': v[6] / max(v[7], 1) / 10})
        print(f'{name} ({args.dtype}): {len(lens)} sequences, {t} tokens, longest {longest}: {ms:.3f} ms, '
              f'{nbytes / ms / 1e6:.0f} GB/s algorithmic', flush=True)


if __name__ == '__main__':
    main()
