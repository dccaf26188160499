#!/usr/bin/env python3
"""What a screen-then-rescore rerank could save: logits of the same (query, chunk) pairs through the float32-equivalent
path (the one whose order counts, reference src/reranker/rerank.py:79-84), and through cheaper passes a screen could
use (the bfloat16 kernels; a float16 forward in PyTorch ops as the yardstick of a single-product fp16 pass).

A screen keeps, per query, the pairs whose cheap logit lies within a margin m of the cheap k-th best; only those are
rescored exactly.  The top k of the exact logits is certain to be among them when m >= 2 e, e = the largest difference
between cheap and exact logit *within the query* (an offset common to all pairs of a query moves no order: it is
removed as the query's median difference).  Prints, for both weight initialisations, the within-query logit spread,
e, and the fraction of pairs the screen would keep at m = 2 e and m = 4 e.

    python scripts/ce_screen_study.py --queries 256 --top-n 50 --out profiles/r03_ce_screen_study.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'domain-rag-system_amd'))

import numpy as np
import torch

from ce_precision import make_pairs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--queries', type=int, default=256)
    ap.add_argument('--top-n', type=int, default=50)
    ap.add_argument('--top-k', type=int, default=5)
    ap.add_argument('--arch', default='minilm-l6')
    ap.add_argument('--seed', type=int, default=5)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    import rag_hip
    from indexing import bert
    dev = torch.device('cuda:0')
    rag_hip.init(0)
    ids, types, pos, cu = make_pairs(args.queries, args.top_n, args.seed)
    n_pairs = len(ids)
    longest = max(len(s) for s in ids)
    t = lambda parts: torch.from_numpy(np.concatenate(parts).astype(np.int32)).to(dev)
    p_ids, p_types, p_pos = t(ids), t(types), t(pos)
    cu_d = torch.from_numpy(cu.astype(np.int32)).to(dev)
    k = args.top_k
    report = {'pairs': n_pairs, 'queries': args.queries, 'top_n': args.top_n, 'top_k': k, 'tokens': int(cu[-1])}

    def padded(model):
        outs = []
        for s0 in range(0, n_pairs, 512):
            it, mt = bert.pad_batch([x.tolist() for x in ids[s0:s0 + 512]], 0, dev)
            tt, _ = bert.pad_batch([x.tolist() for x in types[s0:s0 + 512]], 0, dev)
            outs.append(model(it, tt, mt).float())
        return torch.cat(outs)

    for init in bert.INITS:
        ld = lambda **kw: bert.load_model('random:' + args.arch, 'cross', dev, init=init, **kw)[0]
        with torch.no_grad():
            exact = ld(dtype=torch.float32, native=True, split_gemm=True, split_dtype=torch.float16).forward_packed(
                p_ids, p_types, p_pos, cu_d, longest)
            ref32 = padded(ld(dtype=torch.float32))
            cheap = {'bf16_kernels': ld(dtype=torch.bfloat16, native=True).forward_packed(p_ids, p_types, p_pos, cu_d, longest),
                     'fp16_torch_ops': padded(ld(dtype=torch.float16)),
                     'bf16_torch_ops': padded(ld(dtype=torch.bfloat16))}
        rag_hip.check_faults()
        x = exact.double().cpu().numpy().reshape(args.queries, args.top_n)
        r = ref32.double().cpu().numpy().reshape(args.queries, args.top_n)
        srt = -np.sort(-x, axis=1)
        gap = (srt[:, :k] - srt[:, 1:k + 1]).min(axis=1)
        res = {'exact_vs_torch_fp32_max_abs': float(np.abs(x - r).max()),
               'logit_mean': float(x.mean()), 'logit_std_within_query': float(x.std(axis=1).mean()),
               'logit_range_within_query': float((x.max(axis=1) - x.min(axis=1)).mean()),
               'kth_minus_last_within_query': float((srt[:, k - 1] - srt[:, -1]).mean()),
               'min_gap_top_k_median': float(np.median(gap)), 'min_gap_top_k_p10': float(np.quantile(gap, 0.1))}
        for name, lg in cheap.items():
            c = lg.double().cpu().numpy().reshape(args.queries, args.top_n)
            d = c - x
            resid = d - np.median(d, axis=1, keepdims=True)
            e_abs, e_res = float(np.abs(d).max()), float(np.abs(resid).max())
            cs = -np.sort(-c, axis=1)
            kth = cs[:, k - 1:k]
            row = {'abs_err_max': e_abs, 'abs_err_mean': float(np.abs(d).mean()),
                   'within_query_err_max': e_res, 'within_query_err_p999': float(np.quantile(np.abs(resid), 0.999)),
                   'within_query_err_mean': float(np.abs(resid).mean()),
                   'top_k_lists_equal_to_exact': int((np.argsort(-c, axis=1, kind='stable')[:, :k]
                                                      == np.argsort(-x, axis=1, kind='stable')[:, :k]).all(axis=1).sum())}
            for mult in (2, 4):
                for tag, e in (('abs', e_abs), ('within_query', e_res)):
                    keep = c >= kth - mult * e
                    # does the kept set hold the exact top k of every query?
                    top = np.argsort(-x, axis=1, kind='stable')[:, :k]
                    held = bool(np.take_along_axis(keep, top, axis=1).all())
                    row['kept_frac_m=%de_%s' % (mult, tag)] = float(keep.mean())
                    row['exact_top_k_inside_m=%de_%s' % (mult, tag)] = held
            res[name] = row
        report[init] = res
    print(json.dumps(report, indent=1))
    if args.out:
        with open(args.out, 'w') as f:
            json.dump(report, f, indent=1)


if __name__ == '__main__':
    main()
