#!/usr/bin/env python3
"""Does the screen's error depend on the pair's length?  Cheap (float16) and exact logits of 256 x 50 bench-like pairs;
the within-query residual (cheap - exact less the query's median) by length bin: max, 99.9th percentile, mean.

    python scripts/micro/screen_error_by_length.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'domain-rag-system_amd'))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))

import numpy as np
import torch

from ce_precision import make_pairs


def main():
    import rag_hip
    from indexing import bert
    dev = torch.device('cuda:0')
    rag_hip.init(0)
    q, n = 512, 50
    ids, types, pos, cu = make_pairs(q, n, 5)
    lens = np.diff(cu)
    longest = int(lens.max())
    t = lambda parts: torch.from_numpy(np.concatenate(parts).astype(np.int32)).to(dev)
    p_ids, p_types, p_pos = t(ids), t(types), t(pos)
    cu_d = torch.from_numpy(cu.astype(np.int32)).to(dev)
    for init in bert.INITS:
        ld = lambda **kw: bert.load_model('random:minilm-l6', 'cross', dev, init=init, **kw)[0]
        with torch.no_grad():
            exact = ld(dtype=torch.float32, native=True, split_gemm=True, split_dtype=torch.float16).forward_packed(
                p_ids, p_types, p_pos, cu_d, longest)
            cheap = ld(dtype=torch.float16, native=True).forward_packed(p_ids, p_types, p_pos, cu_d, longest)
        d = (cheap.double() - exact.double()).cpu().numpy().reshape(q, n)
        resid = np.abs(d - np.median(d, axis=1, keepdims=True)).reshape(-1)
        print(init, 'all pairs: max %.3e  p99.9 %.3e  mean %.3e' % (resid.max(), np.quantile(resid, 0.999), resid.mean()))
        for lo, hi in ((0, 40), (40, 60), (60, 80), (80, 100), (100, 130), (130, 999)):
            m = (lens >= lo) & (lens < hi)
            if m.sum():
                r = resid[m]
                print('  length %3d..%3d: %6d pairs  max %.3e  p99.9 %.3e  mean %.3e' % (lo, hi, m.sum(), r.max(), np.quantile(r, 0.999), r.mean()))


if __name__ == '__main__':
    main()
