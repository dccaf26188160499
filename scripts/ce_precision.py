#!/usr/bin/env python3
"""Cross-encoder precision study on the GPU: logits of the same (query, chunk) pairs through
  (a) the bfloat16 librag_hip path (the fast one),
  (b) the float32 path with split-bf16 GEMMs (fp32-equivalent, librag_hip kernels between the GEMMs),
  (c) plain PyTorch float32 ops (what sentence-transformers' CrossEncoder.predict computes,
      reference src/reranker/rerank.py:79-84).
Prints error statistics, per-query logit spread, and how often the top-k order of (a) and (b) equals (c)'s.

    python scripts/ce_precision.py --queries 128 --top-n 50 [--init-std 0.02]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'domain-rag-system_amd'))

import numpy as np
import torch


def make_pairs(n_query, top_n, seed, vocab=30522):
    """Synthetic pairs shaped like the bench workload: [CLS] q [SEP] chunk [SEP]."""
    rng = np.random.default_rng(seed)
    ids, types, pos, cu = [], [], [], [0]
    for _ in range(n_query):
        q = rng.integers(1000, vocab, size=int(rng.integers(8, 25)))
        for _ in range(top_n):
            ln = int(np.clip(np.round(rng.lognormal(np.log(45), 0.6)), 2, 128)) + 2
            c = rng.integers(1000, vocab, size=ln)
            seq = np.concatenate([[101], q, [102], c, [102]])
            ids.append(seq)
            types.append(np.concatenate([np.zeros(len(q) + 2, np.int64), np.ones(ln + 1, np.int64)]))
            pos.append(np.arange(len(seq)))
            cu.append(cu[-1] + len(seq))
    return ids, types, pos, np.asarray(cu, np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--queries', type=int, default=128)
    ap.add_argument('--top-n', type=int, default=50)
    ap.add_argument('--top-k', type=int, default=5)
    ap.add_argument('--arch', default='minilm-l6')
    ap.add_argument('--seed', type=int, default=5)
    ap.add_argument('--init', default=None, help='weight initialisation of the random model (bert.INITS)')
    ap.add_argument('--out', default=None)
    ap.add_argument('--ref64', action='store_true', help='also compare every form with a float64 forward')
    ap.add_argument('--halves', default='fp16', choices=['fp16', 'bf16'], help='16-bit type of the split operands')
    ap.add_argument('--only', default=None, choices=['bf16', 'split'], help='run one path only (for profiling)')
    args = ap.parse_args()
    import rag_hip
    from indexing import bert
    dev = torch.device('cuda:0')
    rag_hip.init(0)
    kw = {} if args.init is None else {'init': args.init}
    ce_bf, _, _ = bert.load_model('random:' + args.arch, 'cross', dev, dtype=torch.bfloat16, native=True, **kw)
    ce_sp, _, _ = bert.load_model('random:' + args.arch, 'cross', dev, dtype=torch.float32, native=True, split_gemm=True,
                                  split_dtype=torch.bfloat16 if args.halves == 'bf16' else torch.float16, **kw)
    ce_32, _, _ = bert.load_model('random:' + args.arch, 'cross', dev, dtype=torch.float32, **kw)

    ids, types, pos, cu = make_pairs(args.queries, args.top_n, args.seed)
    n_pairs = len(ids)
    longest = max(len(s) for s in ids)
    t = lambda parts: torch.from_numpy(np.concatenate(parts).astype(np.int32)).to(dev)
    p_ids, p_types, p_pos = t(ids), t(types), t(pos)
    cu_d = torch.from_numpy(cu.astype(np.int32)).to(dev)

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return out, (time.perf_counter() - t0) / reps * 1e3

    if args.only:
        m = ce_bf if args.only == 'bf16' else ce_sp
        with torch.no_grad():
            _, ms = timed(lambda: m.forward_packed(p_ids, p_types, p_pos, cu_d, longest))
        print(json.dumps({'path': args.only, 'pairs': n_pairs, 'tokens': int(cu[-1]), 'ms': round(ms, 2)}))
        return
    with torch.no_grad():
        lg_bf, ms_bf = timed(lambda: ce_bf.forward_packed(p_ids, p_types, p_pos, cu_d, longest))
        lg_sp, ms_sp = timed(lambda: ce_sp.forward_packed(p_ids, p_types, p_pos, cu_d, longest))
        # plain PyTorch fp32 on padded batches of 512 pairs
        def torch32():
            outs = []
            for s0 in range(0, n_pairs, 512):
                it, mt = bert.pad_batch([x.tolist() for x in ids[s0:s0 + 512]], 0, dev)
                tt, _ = bert.pad_batch([x.tolist() for x in types[s0:s0 + 512]], 0, dev)
                outs.append(ce_32(it, tt, mt))
            return torch.cat(outs)
        lg_32, ms_32 = timed(torch32, reps=1)
    rag_hip.check_faults()
    if args.ref64:
        # float64 forward as the yardstick: errors of the float32 PyTorch forward and of both split forms
        ce_64, _, _ = bert.load_model('random:' + args.arch, 'cross', dev, dtype=torch.float64, **kw)
        ce_sb, _, _ = bert.load_model('random:' + args.arch, 'cross', dev, dtype=torch.float32, native=True, split_gemm=True,
                                      split_dtype=torch.bfloat16, **kw)
        ce_sh, _, _ = bert.load_model('random:' + args.arch, 'cross', dev, dtype=torch.float32, native=True, split_gemm=True,
                                      split_dtype=torch.float16, **kw)
        with torch.no_grad():
            outs = []
            for s0 in range(0, n_pairs, 256):
                it, mt = bert.pad_batch([x.tolist() for x in ids[s0:s0 + 256]], 0, dev)
                tt, _ = bert.pad_batch([x.tolist() for x in types[s0:s0 + 256]], 0, dev)
                h = ce_64.bert(it, tt, mt)
                outs.append(ce_64.classifier(torch.tanh(ce_64.pooler(h[:, 0]))).squeeze(-1))
            r64 = torch.cat(outs)
            sb = ce_sb.forward_packed(p_ids, p_types, p_pos, cu_d, longest).double()
            sh = ce_sh.forward_packed(p_ids, p_types, p_pos, cu_d, longest).double()
        def stat(x):
            e = (x - r64).abs()
            return {'max': float(e.max()), 'p999': float(torch.quantile(e, 0.999)), 'mean': float(e.mean())}
        print(json.dumps({'vs_float64': {'torch_fp32': stat(lg_32.double()), 'split_bf16_halves': stat(sb),
                                         'split_fp16_halves': stat(sh)}}, indent=1))
    a = lg_bf.double().cpu().numpy().reshape(args.queries, args.top_n)
    s = lg_sp.double().cpu().numpy().reshape(args.queries, args.top_n)
    r = lg_32.double().cpu().numpy().reshape(args.queries, args.top_n)
    k = args.top_k

    def order(x):
        return np.argsort(-x, axis=1, kind='stable')[:, :k]

    def gaps(x):
        srt = -np.sort(-x, axis=1)
        return (srt[:, :k] - srt[:, 1:k + 1]).min(axis=1)          # smallest gap among places 1..k+1
    g = gaps(r)
    res = {
        'pairs': n_pairs, 'tokens': int(cu[-1]), 'init': args.init,
        'fp32_logit_mean': float(r.mean()), 'fp32_logit_std_within_query': float(r.std(axis=1).mean()),
        'fp32_logit_range_within_query': float((r.max(axis=1) - r.min(axis=1)).mean()),
        'fp32_min_gap_top_k_median': float(np.median(g)), 'fp32_min_gap_top_k_p10': float(np.quantile(g, 0.1)),
        'bf16_abs_err_max': float(np.abs(a - r).max()), 'bf16_abs_err_mean': float(np.abs(a - r).mean()),
        'split_abs_err_max': float(np.abs(s - r).max()), 'split_abs_err_mean': float(np.abs(s - r).mean()),
        'bf16_top_k_order_equal': int((order(a) == order(r)).all(axis=1).sum()),
        'split_top_k_order_equal': int((order(s) == order(r)).all(axis=1).sum()),
        'queries': args.queries,
        'ms_bf16': round(ms_bf, 2), 'ms_split': round(ms_sp, 2), 'ms_torch_fp32': round(ms_32, 2),
    }
    # pairs the bf16 pass cannot place: within 2*err_max of the k-th / (k+1)-th logit of their query
    delta = 2 * res['bf16_abs_err_max']
    srt = -np.sort(-a, axis=1)
    near = (a >= (srt[:, k - 1:k] - delta))
    res['bf16_pairs_within_2err_of_kth_frac'] = float(near.mean())
    print(json.dumps(res, indent=1))
    if args.out:
        with open(args.out, 'w') as f:
            json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
