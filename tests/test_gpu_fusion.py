"""GPU parity: rank fusion, segmented rerank top-k, list merge and pair packing
(librag_hip via the C ABI) against the golden fixtures and the CPU oracle."""
import json
import os

import numpy as np
import pytest

from oracle.fusion_ref import rrf_keys, weighted_keys
from oracle.rerank_ref import rerank_order_ref

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'reference_pure_python.json')))


def _ids_to_rows(*lists):
    """chunk_id strings -> row numbers; duplicates of one id get distinct rows that share
    a canonical row (what the index-time canon table does)."""
    canon_of = {}
    canon = []
    out = []
    for ids in lists:
        rows = []
        for cid in ids:
            r = len(canon)
            canon_of.setdefault(cid, r)
            canon.append(canon_of[cid])
            rows.append(r)
        out.append(rows)
    return out, np.asarray(canon, np.uint32)


@pytest.mark.parametrize('name', sorted(GOLD['rrf']))
def test_rrf_matches_reference_fixture(name):
    import rag_hip
    case = GOLD['rrf'][name]
    v_ids = [r['chunk_id'] for r in case['vector']]
    b_ids = [r['chunk_id'] for r in case['bm25']]
    (v_rows, b_rows), canon = _ids_to_rows(v_ids, b_ids)
    pos, row, score, cnt = rag_hip.fuse_rrf([v_rows], [b_rows], case['top_k'], canon=canon if len(canon) else None)
    both = case['vector'] + case['bm25']
    exp = case['expected']
    assert int(cnt[0]) == len(exp)
    for j, e in enumerate(exp):
        assert both[pos[0, j]]['chunk_id'] == e['chunk_id']
        assert score[0, j] == e['score']                      # bit-exact fp64
        payload = dict(both[pos[0, j]], score=e['score'], fusion_method='rrf')
        assert payload == e


@pytest.mark.parametrize('name', sorted(GOLD['weighted']))
def test_weighted_matches_reference_fixture(name):
    import rag_hip
    case = GOLD['weighted'][name]
    v_ids = [r['chunk_id'] for r in case['vector']]
    b_ids = [r['chunk_id'] for r in case['bm25']]
    (v_rows, b_rows), canon = _ids_to_rows(v_ids, b_ids)
    v_sc = [r['score'] for r in case['vector']]
    b_sc = [r['score'] for r in case['bm25']]
    pos, row, score, cnt = rag_hip.fuse_weighted([v_rows], [v_sc], [b_rows], [b_sc], case['top_k'],
                                                 vector_weight=case['vector_weight'],
                                                 bm25_weight=case['bm25_weight'],
                                                 canon=canon if len(canon) else None)
    both = case['vector_after'] + case['bm25_after']
    exp = case['expected']
    assert int(cnt[0]) == len(exp)
    for j, e in enumerate(exp):
        assert score[0, j] == e['score']
        payload = dict(both[pos[0, j]], score=e['score'], fusion_method='weighted')
        assert payload == e


@pytest.mark.parametrize('seed,B', [(0, 37), (1, 37), (2, 1), (3, 128), (4, 129), (5, 300)])
def test_fusion_random_batches_match_oracle(seed, B):
    """(up to 128 queries a workgroup of four waves fuses one query, beyond that a wave: both forms, both sides of the switch)"""
    import rag_hip
    rng = np.random.default_rng(seed)
    n_rows = 5000
    canon = np.arange(n_rows, dtype=np.uint32)
    for d in np.sort(rng.choice(np.arange(1, n_rows), 400, replace=False)):
        canon[d] = canon[int(rng.integers(0, d))]              # an earlier row's canonical row
    T = int(rng.integers(1, 60))
    v_rows, b_rows, v_sc, b_sc = [], [], [], []
    for q in range(B):
        nv, nb = int(rng.integers(0, 2 * T + 1)), int(rng.integers(0, 2 * T + 1))
        pool = rng.choice(n_rows, size=3 * T + 5, replace=False)
        v_rows.append(list(rng.choice(pool, nv, replace=False)))
        b_rows.append(list(rng.choice(pool, nb, replace=False)))
        v_sc.append(list(np.sort(rng.random(nv).astype(np.float32))[::-1]))
        b_sc.append(list(np.sort(rng.random(nb) * 30)[::-1]))
    pos, row, score, cnt = rag_hip.fuse_rrf(v_rows, b_rows, T, canon=canon)
    for q in range(B):
        want = rrf_keys([int(canon[r]) for r in v_rows[q]], [int(canon[r]) for r in b_rows[q]], top_k=T)
        assert int(cnt[q]) == len(want)
        assert [(int(pos[q, j]), float(score[q, j])) for j in range(len(want))] == want
        both = v_rows[q] + b_rows[q]
        assert [int(row[q, j]) for j in range(len(want))] == [both[p] for p, _ in want]
    pos, row, score, cnt = rag_hip.fuse_weighted(v_rows, v_sc, b_rows, b_sc, T, 0.7, 0.3, canon=canon)
    for q in range(B):
        want, _, _ = weighted_keys([int(canon[r]) for r in v_rows[q]], [float(s) for s in v_sc[q]],
                                   [int(canon[r]) for r in b_rows[q]], [float(s) for s in b_sc[q]], 0.7, 0.3, T)
        assert int(cnt[q]) == len(want)
        assert [(int(pos[q, j]), float(score[q, j])) for j in range(len(want))] == want


def test_stored_result_scores_decode_as_rrf_sums():
    """The fused scores stored by the reference's evaluation runs are what the kernel emits
    for the corresponding rank pairs."""
    import rag_hip
    stored = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'results_rrf_scores.json')))['scores']
    singles = {1 / (60 + r): (r,) for r in range(1, 101)}
    doubles = {1 / (60 + a) + 1 / (60 + b): (a, b) for a in range(1, 101) for b in range(1, 101)}
    checked = 0
    for vals in stored.values():
        for s in vals:
            ranks = singles.get(s) or doubles.get(s)
            if ranks is None:
                assert s > 2 / 61                      # three or more terms: duplicate chunk ids in a list
                continue
            # build lists where one row sits at those ranks
            v = list(range(1000, 1100))
            b = list(range(2000, 2100))
            v[ranks[0] - 1] = 7
            if len(ranks) == 2:
                b[ranks[1] - 1] = 7
            pos, row, score, cnt = rag_hip.fuse_rrf([v], [b], 200)
            got = {int(r): float(x) for r, x in zip(row[0, :cnt[0]], score[0, :cnt[0]])}
            assert got[7] == s
            checked += 1
    assert checked >= 80


def test_topk_segmented_matches_oracle():
    import rag_hip
    rng = np.random.default_rng(3)
    segs = [0, 50, 50, 51, 81, 200, 230]
    scores = rng.standard_normal(segs[-1]).astype(np.float32)
    scores[60:70] = scores[60]                               # ties keep pool order
    idx, cnt = rag_hip.topk_segmented(scores, segs, 10)
    for q in range(len(segs) - 1):
        want = rerank_order_ref(list(scores[segs[q]:segs[q + 1]]), 10)
        assert int(cnt[q]) == len(want)
        assert list(idx[q, :len(want)]) == want
        assert np.all(idx[q, len(want):] == -1)
    for name, case in GOLD['rerank'].items():
        n = min(case['top_n'], len(case['results']))
        if n == 0:
            continue
        idx, cnt = rag_hip.topk_segmented(np.asarray(case['scores'][:n], np.float32), [0, n], case['top_k'])
        got = [case['results'][i]['chunk_id'] for i in idx[0, :cnt[0]]]
        assert got == [e['chunk_id'] for e in case['expected']], name


def test_merge_topk_matches_sort():
    import torch
    import rag_hip
    rng = np.random.default_rng(5)
    B, L, K = 9, 8, 100
    dev = torch.device('cuda:0')
    for dtype in (np.float32, np.float64):
        score = rng.standard_normal((B, L, K)).astype(dtype)
        score[:, :, ::7] = 0.25                                # plenty of ties
        row = rng.permutation(B * L * K).reshape(B, L, K).astype(np.int64)
        count = rng.integers(0, K + 1, size=(B, L)).astype(np.int32)
        count[0] = 0
        s, r, c = rag_hip.merge_topk_dev(torch.from_numpy(score).to(dev), torch.from_numpy(row).to(dev),
                                         torch.from_numpy(count).to(dev), 100)
        torch.cuda.synchronize()
        s, r, c = s.cpu().numpy(), r.cpu().numpy(), c.cpu().numpy()
        for q in range(B):
            items = [(score[q, l, j], row[q, l, j]) for l in range(L) for j in range(count[q, l])]
            items.sort(key=lambda t: (t[0], t[1]), reverse=True)
            items = items[:100]
            assert c[q] == len(items)
            assert [int(x) for x in r[q, :c[q]]] == [int(t[1]) for t in items]
            assert np.array_equal(s[q, :c[q]], np.asarray([t[0] for t in items], dtype))


@pytest.mark.parametrize('k', [1, 10, 16, 100])
def test_merge_lists_with_a_narrow_score_range(k):
    """All scores positive and close together: the keys share their high bits, so the radix
    select starts below the top byte."""
    import torch
    import rag_hip
    rng = np.random.default_rng(k)
    B, L = 5, 8
    dev = torch.device('cuda:0')
    for dtype, lo, span in ((np.float32, 0.05, 0.1), (np.float64, 11.0, 3.0), (np.float32, 0.7, 1e-5)):
        score = (lo + span * rng.random((B, L, k))).astype(dtype)
        row = rng.permutation(B * L * k).reshape(B, L, k).astype(np.int64)
        s, r, c = rag_hip.merge_topk_dev(torch.from_numpy(score).to(dev), torch.from_numpy(row).to(dev), None, k)
        torch.cuda.synchronize()
        for q in range(B):
            items = sorted(zip(score[q].ravel().tolist(), row[q].ravel().tolist()), reverse=True)[:k]
            assert int(c[q]) == k
            assert r[q].cpu().tolist() == [int(t[1]) for t in items]
            assert np.array_equal(s[q].cpu().numpy(), np.asarray([t[0] for t in items], dtype))


def test_ce_pack_layout():
    import torch
    import rag_hip
    dev = torch.device('cuda:0')
    q_tok = torch.tensor([11, 12, 13, 21, 22], dtype=torch.int32, device=dev)
    q_off = torch.tensor([0, 3, 5], dtype=torch.int64, device=dev)
    c_tok = torch.tensor([31, 32, 33, 34, 41, 51, 52, 53, 54, 55, 56, 57, 58, 59], dtype=torch.int32, device=dev)
    c_off = torch.tensor([0, 4, 5, 14], dtype=torch.int64, device=dev)
    pair_q = torch.tensor([0, 1, 1, 0], dtype=torch.int32, device=dev)
    pair_r = torch.tensor([100, 102, -1, 101], dtype=torch.int64, device=dev)
    slot = torch.tensor([3, 0, 1, 2], dtype=torch.int32, device=dev)
    ln = rag_hip.ce_pair_len_dev(q_off, c_off, 100, pair_q, pair_r, 12).cpu().tolist()
    assert ln == [3 + 4 + 3, 12, 0, 3 + 1 + 3]
    ids, types, mask = rag_hip.ce_pack_dev(q_tok, q_off, c_tok, c_off, 100, pair_q, pair_r, slot, 4, 12,
                                           cls_id=101, sep_id=102, pad_id=0)
    torch.cuda.synchronize()
    ids, types, mask = ids.cpu().tolist(), types.cpu().tolist(), mask.cpu().tolist()
    assert ids[3] == [101, 11, 12, 13, 102, 31, 32, 33, 34, 102, 0, 0]
    assert types[3] == [0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 0, 0]
    assert mask[3] == [1] * 10 + [0, 0]
    assert ids[0] == [101, 21, 22, 102, 51, 52, 53, 54, 55, 56, 57, 102]      # chunk truncated to fit
    assert mask[0] == [1] * 12
    assert ids[1] == [0] * 12 and mask[1] == [0] * 12
    assert ids[2] == [101, 11, 12, 13, 102, 41, 102, 0, 0, 0, 0, 0]
