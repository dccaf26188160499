"""GPU: indices read from device data are range-checked inside the kernels; a bad value sets a
flag (rag_fault_flags) and the kernel carries on with a safe substitute instead of touching
memory outside its arrays."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_flags_start_clear_and_clear_on_read():
    import rag_hip
    rag_hip.fault_flags()
    assert rag_hip.fault_flags(clear=False) == 0


def test_fusion_row_outside_canon_is_flagged_not_read():
    import rag_hip
    canon = _dev(np.arange(10, dtype=np.uint32))
    vec = _dev(np.array([[3, 999999999, 5]], np.int64))
    bm = _dev(np.array([[5, -7, 3]], np.int64))
    cnt = _dev(np.array([3], np.int32))
    pos, row, score, n = rag_hip.fuse_rrf_dev(vec, cnt, bm, cnt, 6, canon=canon)
    assert rag_hip.fault_flags() == 32
    # the in-range ids still fuse: 3 and 5 appear in both lists
    got = dict(zip(row[0, :int(n[0])].tolist(), score[0, :int(n[0])].tolist()))
    assert got[3] == 1 / 61 + 1 / 63 and got[5] == 1 / 63 + 1 / 61
    assert int(n[0]) == 4


def test_cross_encoder_pair_out_of_range_is_padded_and_flagged():
    import torch
    import rag_hip
    q_tok = _dev(np.array([11, 12, 13], np.int32))
    q_off = _dev(np.array([0, 3], np.int64))
    c_tok = _dev(np.array([21, 22, 23, 24], np.int32))
    c_off = _dev(np.array([0, 2, 4], np.int64))
    pair_q = _dev(np.array([0, 0, 5, 0], np.int32))
    pair_r = _dev(np.array([1, 77, 0, -1], np.int64))          # ok, row too large, query too large, empty
    ln = rag_hip.ce_pair_len_dev(q_off, c_off, 0, pair_q, pair_r, 16).cpu().tolist()
    assert ln == [8, 0, 0, 0]
    assert rag_hip.fault_flags() == 64
    ids, types, mask = rag_hip.ce_pack_dev(q_tok, q_off, c_tok, c_off, 0, pair_q, pair_r, None, 4, 8)
    assert rag_hip.fault_flags() == 64
    assert ids[0].tolist() == [101, 11, 12, 13, 102, 23, 24, 102]
    assert int(mask[1:].sum()) == 0 and bool((ids[1:] == 0).all())
    # a slot beyond the output is skipped
    slot = _dev(np.array([0, 9, 1, 2], np.int32))
    rag_hip.ce_pack_dev(q_tok, q_off, c_tok, c_off, 0, pair_q, torch.zeros_like(pair_r), slot, 4, 8)
    assert rag_hip.fault_flags() == 64


def test_bm25_term_beyond_vocabulary_counts_as_unknown():
    import rag_hip
    from indexing.bm25_indexer import build_arrays
    term = np.array([0, 1, 1, 2], np.int64)
    doc = np.array([0, 0, 1, 2], np.int64)
    arr = build_arrays(term, doc, np.array([2, 1, 1], np.int64), 3, 1.5, 0.75)
    bm = rag_hip.Bm25Index(arr.indptr, arr.rows, arr.impacts(), arr.idf, 3)
    good = bm.search([[1]], 3)
    bad = bm.search([[1, 12345]], 3)
    assert rag_hip.fault_flags() == 16
    assert np.array_equal(good[1], bad[1]) and np.array_equal(good[0], bad[0])


def test_segmented_topk_bad_offsets_give_empty_pools():
    import rag_hip
    scores = _dev(np.array([0.5, 0.25, 0.75], np.float32))
    seg = _dev(np.array([0, 3, 2, 40], np.int32))               # pool 1 descending, pool 2 past the end
    idx, cnt = rag_hip.topk_segmented_dev(scores, seg, 2)
    assert rag_hip.fault_flags() == 128
    assert cnt.tolist() == [2, 0, 0] and idx[0].tolist() == [2, 0]
