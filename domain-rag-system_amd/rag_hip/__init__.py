"""ctypes binding of ``librag_hip.so`` (declared in ``include/rag_hip.h``).

The library is the only compute backend of this package: when it is missing, or no
gfx950 GPU is usable, the entry points raise.  There is no CPU fallback.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RAG_HIP_LIB') or os.path.join(_HERE, 'librag_hip.so')     # the override is for A/B builds of a kernel

METRIC_COSINE, METRIC_DOT, METRIC_EUCLID = 0, 1, 2
MEM_HOST_COPY, MEM_DEVICE_COPY, MEM_DEVICE_BORROW = 0, 1, 2
MAX_K = 1024
MAX_FUSE_LIST = 512
DTYPE_F32, DTYPE_BF16, DTYPE_F16 = 0, 1, 2
ENC_MAX_SEQ = 512

_METRICS = {'cosine': METRIC_COSINE, 'dot': METRIC_DOT, 'euclidean': METRIC_EUCLID, 'euclid': METRIC_EUCLID}


class RagHipError(RuntimeError):
    pass


_vp, _i32, _i64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double

# name -> (restype, argtypes); every symbol of include/rag_hip.h
SIGNATURES = {
    'rag_init': (_i32, [_i32]),
    'rag_last_error': (C.c_char_p, []),
    'rag_version': (_i32, []),
    'rag_fault_flags': (_i32, [C.POINTER(C.c_uint32), _i32]),
    'rag_dense_create': (_i32, [_vp, _i64, _i32, _i32, _i32, _i64, C.POINTER(_vp)]),
    'rag_dense_destroy': (_i32, [_vp]),
    'rag_dense_rows': (_i64, [_vp]),
    'rag_dense_enable_split': (_i32, [_vp, _i32]),
    'rag_dense_profile': (_i32, [_vp, _i32]),
    'rag_dense_profile_read': (_i32, [_vp, _vp, _i32, C.POINTER(_i32)]),
    'rag_dense_search_dev': (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'rag_dense_search': (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    'rag_dense_rescore_dev': (_i32, [_vp, _vp, _i32, _vp, _i32, _vp, _vp]),
    'rag_dense_rescore': (_i32, [_vp, _vp, _i32, _vp, _i32, _vp]),
    'rag_bm25_create': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i64, C.POINTER(_vp)]),
    'rag_bm25_destroy': (_i32, [_vp]),
    'rag_bm25_docs': (_i64, [_vp]),
    'rag_bm25_search_dev': (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    'rag_bm25_search': (_i32, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    'rag_bm25_search_static_dev': (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    'rag_bm25_search_static_capacity': (_i64, []),
    'rag_bm25_scores': (_i32, [_vp, _vp, _i32, _vp]),
    'rag_merge_topk_f32_dev': (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    'rag_merge_topk_f64_dev': (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    'rag_fuse_rrf_dev': (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _f64, _vp, _vp, _vp, _vp,
                                _vp]),
    'rag_fuse_weighted_dev': (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _f64, _f64,
                                     _vp, _vp, _vp, _vp, _vp]),
    'rag_fuse_rrf': (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _f64, _vp, _vp, _vp, _vp]),
    'rag_fuse_weighted': (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _i64, _i32, _i32, _f64, _f64,
                                 _vp, _vp, _vp, _vp]),
    'rag_topk_segmented_dev': (_i32, [_vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp]),
    'rag_topk_segmented': (_i32, [_vp, _vp, _i32, _i32, _vp, _vp]),
    'rag_ce_pack_dev': (_i32, [_vp, _vp, _i32, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                               _i32, _vp, _vp, _vp, _vp]),
    'rag_ce_pair_len_dev': (_i32, [_vp, _i32, _vp, _i64, _i64, _vp, _vp, _i32, _i32, _vp, _vp]),
    'rag_enc_attention_dev': (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    'rag_enc_attention_packed_dev': (_i32, [_vp, _vp, _i32, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
    'rag_enc_attention_packed_first_dev': (_i32, [_vp, _vp, _i32, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
    'rag_enc_attention_split_dev': (_i32, [_vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    'rag_enc_attention_split_hl_dev': (_i32, [_vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    'rag_enc_embed_layernorm_dev': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _i64, _i32, _i32, _i32,
                                           _i32, _i32, _vp, _vp]),
    'rag_enc_embed_layernorm_split_dev': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _i64, _i32, _i32, _i32,
                                                 _i32, _vp, _i32, _vp, _vp]),
    'rag_enc_mlp_dev': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _i64, _i32, _i32, _i32, _vp, _vp]),
    'rag_filter_mask_dev': (_i32, [_vp, _i32, _i64, _vp, _vp, _vp]),
    'rag_dense_reserve': (_i32, [_vp, _i32, _i32]),
    'rag_bm25_reserve': (_i32, [_vp, _i32, _i32, _i32]),
    'rag_bm25_tokenize_create': (_i32, [C.c_char_p, _vp, _vp, _i64, C.c_char_p, _i32, C.POINTER(_vp)]),
    'rag_bm25_tokenize_sizes': (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    'rag_bm25_tokenize_fetch': (_i32, [_vp, _vp, _vp, _vp, _vp]),
    'rag_bm25_tokenize_destroy': (None, [_vp]),
    'rag_bm25_build_dev': (_i32, [_vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, C.POINTER(_i64), _vp]),
    'rag_bm25_impacts_dev': (_i32, [_vp, _vp, _i64, _vp, _i64, C.c_double, C.c_double, C.c_double, _vp, _vp]),
    'rag_enc_linear384_dev': (_i32, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, C.c_float, _i32, _vp, _vp]),
    'rag_enc_qkv_attention_tiles': (_i32, [_vp, _i32, _vp, _vp]),
    'rag_enc_qkv_attention_pack': (_i32, [_vp, _i32, _vp, _vp, _vp]),
    'rag_enc_qkv_attention_dev': (_i32, [_vp, _vp, _vp, _vp, _i32, _i64, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    'rag_enc_qkv_attention_first_dev': (_i32, [_vp, _vp, _vp, _vp, _i32, _i64, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    'rag_enc_mlp_profile': (_i32, [_i32]),
    'rag_enc_mlp_profile_read': (_i32, [_vp, _vp, _i32, C.POINTER(_i32)]),
    'rag_enc_attn_out_mlp_dev': (_i32, [_vp] * 12 + [C.c_float, _i64, _i32, _i32, _i32, _vp, _vp]),
    'rag_enc_norm_mlp_dev': (_i32, [_vp] * 10 + [C.c_float, _i64, _i32, _i32, _i32, _vp, _vp]),
    'rag_enc_gelu_dev': (_i32, [_vp, _i64, _i32, _vp]),
    'rag_enc_split_bf16x3_dev': (_i32, [_vp, _i64, _i32, _i32, _vp, _vp]),
    'rag_enc_bias_act_split_bf16x3_dev': (_i32, [_vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp]),
    'rag_enc_bias_add_layernorm_split_dev': (_i32, [_vp, _vp, _vp, _vp, _vp, C.c_float, _i64, _i32, _i32, _vp, _vp, _vp]),
    'rag_enc_mean_pool_dev': (_i32, [_vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp]),
    'rag_ce_pack_packed_dev': (_i32, [_vp, _vp, _i32, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _i32, _i64, _i32, _i32,
                                      _i32, _vp, _vp, _vp, _vp]),
    'rag_enc_add_layernorm_dev': (_i32, [_vp, _vp, _vp, _vp, C.c_float, _i64, _i32, _i32, _vp, _vp]),
    'rag_enc_linear_ln_split_dev': (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _i64, _i32, _i32, _vp, _vp, _vp]),
    'rag_enc_linear_ln_split_chain_dev': (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _i64, _i32, _i32, _vp, _vp, _vp]),
    'rag_enc_linear_ln_split_pack_dev': (_i32, [_vp, _i32, _i32, _vp, _vp, _vp]),
    'rag_enc_linear_ln_split_profile': (_i32, [_i32]),
    'rag_enc_linear_ln_split_profile_read': (_i32, [_vp, _vp, _vp, _i32, C.POINTER(_i32)]),
}

_lib = None
_lock = threading.Lock()
_inited_devices = set()


def load_library():
    """dlopen the library and declare every prototype.  Does not touch the GPU."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RagHipError(
                    f"{LIB_PATH} is missing: build it with `make -C domain-rag-system_amd/csrc` "
                    "(or __graft_entry__.build()); this package has no CPU fallback")
            # PyTorch ships its own copy of the HIP runtime (same SONAME as /opt/rocm's).  Two
            # copies in one process fight over the device, so when torch is installed its copy
            # must be the one already loaded when this library resolves libamdhip64.so.7.
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def last_error():
    return load_library().rag_last_error().decode('utf-8', 'replace')


def check(rc, what):
    if rc != 0:
        raise RagHipError(f"{what} failed (code {rc}): {last_error()}")


def init(device=0):
    """Bind the calling thread to `device` and verify it is a gfx950 part."""
    lib = load_library()
    check(lib.rag_init(int(device)), 'rag_init')
    _inited_devices.add(int(device))
    return lib


def lib():
    l = load_library()
    if not _inited_devices:
        init(0)
    return l


def metric_code(name):
    try:
        return _METRICS[str(name).lower()]
    except KeyError:
        raise ValueError(f"unknown distance metric {name!r}")


def _np(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, a.ctypes.data


def _ptr(t, dtype=None):
    """Device pointer of a torch tensor (or None); the kernels read raw memory, so layout,
    element type and placement are checked here."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("rag_hip device entry points need tensors on the GPU")
    if not t.is_contiguous():
        raise ValueError("device tensors passed to rag_hip must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"rag_hip expected a {dtype} tensor, got {t.dtype}")
    return t.data_ptr()


FAULT_NAMES = {1: 'dense candidate list overflow', 2: 'merge selection short', 4: 'bm25 tile list overflow',
               8: 'bm25 posting outside its tile', 16: 'bm25 term id beyond the vocabulary',
               32: 'fusion row outside the canon table', 64: 'cross-encoder pair out of range',
               128: 'segment offsets out of range', 256: 'token, position or type id outside its embedding table'}


def fault_flags(clear=True):
    """RAG_FAULT_* bits the kernels raised on the current device since the last clear
    (synchronises the device); 0 means every device-side check held."""
    out = C.c_uint32(0)
    check(lib().rag_fault_flags(C.byref(out), 1 if clear else 0), 'rag_fault_flags')
    return int(out.value)


def check_faults():
    """Raise if any device-side check failed since the last clear."""
    bits = fault_flags(clear=True)
    if bits:
        names = [n for b, n in FAULT_NAMES.items() if bits & b]
        raise RagHipError(f"device-side consistency checks failed (flags {bits:#x}): " + '; '.join(names))


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


# ---------------------------------------------------------------------------------------
class DenseIndex:
    """Resident [N x D] fp32 matrix with exact top-k search (rag_dense_* entry points)."""

    def __init__(self, rows, metric='cosine', row_base=0, borrow=False, split='auto'):
        """split: run searches as three bf16 matrix-core products of split operands (scores within ~1e-7 of
        the fp32 products for unit vectors, a fifth of the matrix time, a second copy of the rows in HBM);
        'auto' = whenever dim is a multiple of 32, False = exact fp32 FMA chains."""
        l = lib()
        self.metric = metric_code(metric)
        self._keep = None
        h = _vp()
        if hasattr(rows, 'data_ptr'):            # torch tensor on the GPU
            import torch
            if rows.dtype != torch.float32 or rows.dim() != 2 or not rows.is_cuda:
                raise ValueError("device rows must be a 2-d float32 CUDA tensor")
            rows = rows.contiguous()
            n, d = rows.shape
            kind = MEM_DEVICE_BORROW if borrow else MEM_DEVICE_COPY
            if borrow:
                self._keep = rows
            torch.cuda.synchronize()
            check(l.rag_dense_create(rows.data_ptr(), n, d, self.metric, kind, int(row_base), C.byref(h)),
                  'rag_dense_create')
        else:
            arr, p = _np(rows, np.float32)
            if arr.ndim != 2:
                raise ValueError("rows must be 2-d")
            n, d = arr.shape
            check(l.rag_dense_create(p, n, d, self.metric, MEM_HOST_COPY, int(row_base), C.byref(h)),
                  'rag_dense_create')
        self.n_rows, self.dim, self.row_base = int(n), int(d), int(row_base)
        self._h = h
        self.split = False
        if split is True and d % 32 != 0:
            raise ValueError("split searches need a dimension that is a multiple of 32")
        if split is True or (split == 'auto' and d % 32 == 0 and not os.environ.get('RAG_DENSE_EXACT')):
            self.set_split(True)

    def set_split(self, enable):
        check(lib().rag_dense_enable_split(self._h, 1 if enable else 0), 'rag_dense_enable_split')
        self.split = bool(enable)

    def close(self):
        if getattr(self, '_h', None):
            load_library().rag_dense_destroy(self._h)
            self._h = None

    __del__ = close

    def search(self, queries, k, row_mask=None):
        """Host arrays in, host arrays out: (scores [B,k] f32, rows [B,k] i64, counts [B] i32)."""
        q, qp = _np(queries, np.float32)
        if q.ndim == 1:
            q = q[None, :]
        if q.shape[1] != self.dim:
            raise ValueError(f"query dim {q.shape[1]} != index dim {self.dim}")
        b = q.shape[0]
        score = np.zeros((b, k), np.float32)
        row = np.full((b, k), -1, np.int64)
        cnt = np.zeros(b, np.int32)
        mp = None
        if row_mask is not None:
            m, mp = _np(row_mask, np.uint8)
            if m.shape != (self.n_rows,):
                raise ValueError("row_mask must have one byte per local row")
        check(lib().rag_dense_search(self._h, q.ctypes.data, b, int(k), mp, score.ctypes.data, row.ctypes.data,
                                     cnt.ctypes.data), 'rag_dense_search')
        return score, row, cnt

    def reserve(self, max_queries, max_k):
        """Size the work buffers now: searches within these bounds then never allocate (rag_dense_reserve)."""
        check(lib().rag_dense_reserve(self._h, int(max_queries), min(int(max_k), MAX_K)), 'rag_dense_reserve')

    def search_dev(self, queries, k, row_mask=None, out=None):
        """torch CUDA tensors in/out, asynchronous on the current stream."""
        import torch
        b = queries.shape[0]
        if queries.dtype != torch.float32 or queries.shape[1] != self.dim:
            raise ValueError("queries must be float32 [B, dim]")
        if out is None:
            out = (torch.empty((b, k), dtype=torch.float32, device=queries.device),
                   torch.empty((b, k), dtype=torch.int64, device=queries.device),
                   torch.empty((b,), dtype=torch.int32, device=queries.device))
        score, row, cnt = out
        if row_mask is not None and row_mask.shape != (self.n_rows,):
            raise ValueError("row_mask must have one byte per local row")
        if tuple(score.shape) != (b, k) or tuple(row.shape) != (b, k) or tuple(cnt.shape) != (b,):
            raise ValueError("output tensors must be [B, k], [B, k], [B]")
        check(lib().rag_dense_search_dev(self._h, _ptr(queries, torch.float32), b, int(k), _ptr(row_mask, torch.uint8),
                                         _ptr(score, torch.float32), _ptr(row, torch.int64), _ptr(cnt, torch.int32),
                                         _stream()), 'rag_dense_search_dev')
        return score, row, cnt

    def profile(self, enable=True):
        check(lib().rag_dense_profile(self._h, 1 if enable else 0), 'rag_dense_profile')

    def profile_read(self):
        """Durations (ms) of the scan kernels launched since the last read."""
        buf = np.zeros(256, np.float32)
        n = _i32(0)
        check(lib().rag_dense_profile_read(self._h, buf.ctypes.data, 256, C.byref(n)), 'rag_dense_profile_read')
        return buf[:n.value].copy()

    def rescore(self, queries, cand_rows):
        q, _ = _np(queries, np.float32)
        if q.ndim == 1:
            q = q[None, :]
        c, _ = _np(cand_rows, np.int64)
        c = c.reshape(q.shape[0], -1)
        out = np.zeros(c.shape, np.float32)
        check(lib().rag_dense_rescore(self._h, q.ctypes.data, q.shape[0], c.ctypes.data, c.shape[1], out.ctypes.data),
              'rag_dense_rescore')
        return out

    def rescore_dev(self, queries, cand_rows):
        import torch
        b, m = cand_rows.shape
        if tuple(queries.shape) != (b, self.dim):
            raise ValueError("queries must be float32 [B, dim] with one row per candidate list")
        out = torch.empty((b, m), dtype=torch.float32, device=queries.device)
        check(lib().rag_dense_rescore_dev(self._h, _ptr(queries, torch.float32), b, _ptr(cand_rows, torch.int64), m,
                                          _ptr(out), _stream()), 'rag_dense_rescore_dev')
        return out


FILTER_ANY, FILTER_NONE, FILTER_MISSING = -1, -2, -3


class RowFilter:
    """Per-row categorical codes resident on the device; ``mask_dev`` turns an exact-match filter
    into the row mask the dense search takes (rag_filter_mask_dev).  ``columns`` maps a key to an
    int32 code per row (>= 0; FILTER_MISSING for rows without the key)."""

    MAX_FIELDS = 8

    def __init__(self, columns, device='cuda'):
        import torch
        self.keys = list(columns)
        if len(self.keys) > self.MAX_FIELDS:
            raise ValueError(f"at most {self.MAX_FIELDS} filterable keys")
        cols = [np.ascontiguousarray(columns[k], np.int32) for k in self.keys]
        self.n_rows = int(cols[0].shape[0]) if cols else 0
        if any(c.shape != (self.n_rows,) for c in cols):
            raise ValueError("every code column needs one entry per row")
        stacked = np.stack(cols) if cols else np.zeros((0, 0), np.int32)
        self.codes = torch.from_numpy(stacked).to(device)

    def mask_dev(self, want):
        """want: key -> wanted code (FILTER_NONE when the value occurs in no row); keys left out are
        unconstrained.  Returns a uint8 [n_rows] tensor on the device."""
        import torch
        unknown = [k for k in want if k not in self.keys]
        if unknown:
            raise KeyError(f"no code column for {unknown}")
        w = np.array([int(want.get(k, FILTER_ANY)) for k in self.keys], np.int32)
        mask = torch.empty((self.n_rows,), dtype=torch.uint8, device=self.codes.device)
        check(lib().rag_filter_mask_dev(_ptr(self.codes, torch.int32) if self.keys else None, len(self.keys), self.n_rows,
                                        w.ctypes.data if self.keys else None, _ptr(mask), _stream()), 'rag_filter_mask_dev')
        return mask


def bm25_tokenize(docs, pretokenized, short_keys=''):
    """Native tokenizer + vocabulary of the BM25 index build (rag_bm25_tokenize_*; host code, no GPU).
    docs: list of bytes objects, one per document: the document's ASCII text, or, where pretokenized[d] is
    true, its tokens (UTF-8) joined by b'\\n'.  short_keys: the one-character lower-cased abbreviation keys as
    one str.  Returns (term_ids int32 [n_tokens], doc_len int64 [n_docs], vocabulary: list of str in id order)."""
    n_docs = len(docs)
    doc_off = np.zeros(n_docs + 1, np.int64)
    if n_docs:
        np.cumsum(np.fromiter((len(d) for d in docs), np.int64, n_docs), out=doc_off[1:])
    text = b''.join(docs)
    flags = np.ascontiguousarray(np.asarray(pretokenized, dtype=np.uint8)) if n_docs else np.zeros(0, np.uint8)
    if flags.shape[0] != n_docs:
        raise ValueError("pretokenized must have one entry per document")
    keys = short_keys.encode('ascii', 'ignore')
    h = _vp()
    check(load_library().rag_bm25_tokenize_create(text, doc_off.ctypes.data, flags.ctypes.data if n_docs else None, n_docs, keys,
                                         len(keys), C.byref(h)), 'rag_bm25_tokenize_create')
    try:
        n_tok, n_voc, n_bytes = _i64(0), _i64(0), _i64(0)
        check(load_library().rag_bm25_tokenize_sizes(h, C.byref(n_tok), C.byref(n_voc), C.byref(n_bytes)), 'rag_bm25_tokenize_sizes')
        term_ids = np.empty(n_tok.value, np.int32)
        doc_len = np.empty(n_docs, np.int64)
        vbytes = np.empty(max(n_bytes.value, 1), np.uint8)
        voff = np.empty(n_voc.value + 1, np.int64)
        check(load_library().rag_bm25_tokenize_fetch(h, term_ids.ctypes.data, doc_len.ctypes.data, vbytes.ctypes.data,
                                            voff.ctypes.data), 'rag_bm25_tokenize_fetch')
    finally:
        load_library().rag_bm25_tokenize_destroy(h)
    raw = vbytes.tobytes()
    vocab = [raw[voff[i]:voff[i + 1]].decode('utf-8') for i in range(n_voc.value)]
    return term_ids, doc_len, vocab


def bm25_build_dev(term, doc_off, vocab):
    """Tokenised documents on the device -> term-major CSR on the device (rag_bm25_build_dev).
    term int32 [n_tokens] (document-major, negative ids ignored), doc_off int64 [n_docs+1];
    returns (indptr int64 [vocab+1], rows uint32-as-int32-bits [nnz], tf [nnz]) as torch tensors."""
    import torch
    n_tokens, n_docs = int(term.shape[0]), int(doc_off.shape[0]) - 1
    indptr = torch.empty((int(vocab) + 1,), dtype=torch.int64, device=term.device)
    rows = torch.empty((max(n_tokens, 1),), dtype=torch.int32, device=term.device)
    tf = torch.empty((max(n_tokens, 1),), dtype=torch.int32, device=term.device)
    nnz = _i64(0)
    check(lib().rag_bm25_build_dev(_ptr(term, torch.int32), n_tokens, _ptr(doc_off, torch.int64), n_docs, int(vocab),
                                   n_tokens, _ptr(indptr), _ptr(rows), _ptr(tf), C.byref(nnz), _stream()),
          'rag_bm25_build_dev')
    return indptr, rows[:nnz.value], tf[:nnz.value]


def bm25_impacts_dev(rows, tf, doc_len, avgdl, k1, b):
    """fp64 impact of every posting on the device (rag_bm25_impacts_dev); rows / tf as returned by
    bm25_build_dev, doc_len int64 [n_docs]."""
    import torch
    out = torch.empty((int(rows.shape[0]),), dtype=torch.float64, device=rows.device)
    check(lib().rag_bm25_impacts_dev(_ptr(rows, torch.int32), _ptr(tf, torch.int32), int(rows.shape[0]),
                                     _ptr(doc_len, torch.int64), int(doc_len.shape[0]), float(avgdl), float(k1), float(b),
                                     _ptr(out), _stream()), 'rag_bm25_impacts_dev')
    return out


class Bm25Index:
    """Term-major CSR postings with fp64 impacts (rag_bm25_* entry points)."""

    def __init__(self, indptr, rows, impact, idf, n_docs, row_base=0):
        l = lib()
        self._arrays = [np.ascontiguousarray(indptr, np.int64), np.ascontiguousarray(rows, np.uint32),
                        np.ascontiguousarray(impact, np.float64), np.ascontiguousarray(idf, np.float64)]
        ip, rw, im, df = self._arrays
        vocab = ip.shape[0] - 1
        if df.shape[0] != vocab or rw.shape[0] != im.shape[0] or (vocab >= 0 and ip[-1] != rw.shape[0]):
            raise ValueError("inconsistent CSR arrays")
        h = _vp()
        check(l.rag_bm25_create(ip.ctypes.data, rw.ctypes.data, im.ctypes.data, df.ctypes.data, int(n_docs), int(vocab),
                                MEM_HOST_COPY, int(row_base), C.byref(h)), 'rag_bm25_create')
        self._arrays = None
        self._h = h
        self.n_docs, self.vocab, self.row_base = int(n_docs), int(vocab), int(row_base)

    def close(self):
        if getattr(self, '_h', None):
            load_library().rag_bm25_destroy(self._h)
            self._h = None

    __del__ = close

    @staticmethod
    def flatten(queries):
        """list of term-id lists -> (qterms int32, qoff int32)."""
        off = np.zeros(len(queries) + 1, np.int32)
        for i, q in enumerate(queries):
            off[i + 1] = off[i] + len(q)
        terms = np.zeros(int(off[-1]), np.int32)
        for i, q in enumerate(queries):
            terms[off[i]:off[i + 1]] = q
        return terms, off

    def search(self, queries, k):
        terms, off = self.flatten(queries)
        b = len(queries)
        score = np.zeros((b, k), np.float64)
        row = np.full((b, k), -1, np.int64)
        cnt = np.zeros(b, np.int32)
        check(lib().rag_bm25_search(self._h, terms.ctypes.data, off.ctypes.data, b, int(k), score.ctypes.data,
                                    row.ctypes.data, cnt.ctypes.data), 'rag_bm25_search')
        return score, row, cnt

    def reserve(self, max_queries, max_terms_total, max_k):
        """Size the work buffers now: searches within these bounds then never allocate (rag_bm25_reserve)."""
        check(lib().rag_bm25_reserve(self._h, int(max_queries), int(max_terms_total), min(int(max_k), MAX_K)), 'rag_bm25_reserve')

    def search_dev(self, qterms, qoff, n_terms_total, k, out=None):
        import torch
        b = qoff.shape[0] - 1
        if out is None:
            out = (torch.empty((b, k), dtype=torch.float64, device=qoff.device),
                   torch.empty((b, k), dtype=torch.int64, device=qoff.device),
                   torch.empty((b,), dtype=torch.int32, device=qoff.device))
        score, row, cnt = out
        if int(n_terms_total) > qterms.shape[0]:
            raise ValueError("n_terms_total exceeds the term array")
        if tuple(score.shape) != (b, k) or tuple(row.shape) != (b, k) or tuple(cnt.shape) != (b,):
            raise ValueError("output tensors must be [B, k], [B, k], [B]")
        check(lib().rag_bm25_search_dev(self._h, _ptr(qterms, torch.int32), _ptr(qoff, torch.int32), b,
                                        int(n_terms_total), int(k), _ptr(score, torch.float64), _ptr(row, torch.int64),
                                        _ptr(cnt, torch.int32), _stream()), 'rag_bm25_search_dev')
        return score, row, cnt

    def search_static_dev(self, qterms, qoff, n_terms_total, k, out):
        """search_dev without any device allocation (rag_bm25_search_static_dev): the work buffers come from a
        static device array, so `out` = (scores [B,k] f64, rows [B,k] i64, counts [B] i32) must be given and
        sizes beyond the array's capacity raise (RAG_ELIMIT).  Needs rag_init on the device; calls from different streams
        are chained on the device (they share the GPU's one array)."""
        import torch
        b = qoff.shape[0] - 1
        score, row, cnt = out
        if int(n_terms_total) > qterms.shape[0]:
            raise ValueError("n_terms_total exceeds the term array")
        if tuple(score.shape) != (b, k) or tuple(row.shape) != (b, k) or tuple(cnt.shape) != (b,):
            raise ValueError("output tensors must be [B, k], [B, k], [B]")
        check(lib().rag_bm25_search_static_dev(self._h, _ptr(qterms, torch.int32), _ptr(qoff, torch.int32), b,
                                               int(n_terms_total), int(k), _ptr(score, torch.float64),
                                               _ptr(row, torch.int64), _ptr(cnt, torch.int32), _stream()),
              'rag_bm25_search_static_dev')
        return score, row, cnt

    def scores(self, query_terms):
        t = np.ascontiguousarray(query_terms, np.int32)
        out = np.zeros(self.n_docs, np.float64)
        check(lib().rag_bm25_scores(self._h, t.ctypes.data, int(t.shape[0]), out.ctypes.data), 'rag_bm25_scores')
        return out


# ---------------------------------------------------------------------------------------
def _pad_lists(lists, dtype, fill):
    b = len(lists)
    stride = max([len(x) for x in lists] + [0])
    arr = np.full((b, max(stride, 1)), fill, dtype)
    cnt = np.zeros(b, np.int32)
    for i, x in enumerate(lists):
        arr[i, :len(x)] = x
        cnt[i] = len(x)
    return arr, cnt, stride


def fuse_rrf(vec_rows, bm_rows, top_k, canon=None, k_const=60.0):
    """Lists of row-id lists (one per query) -> (pos [B,top_k], row, score f64, count)."""
    b = len(vec_rows)
    v, vc, vs = _pad_lists(vec_rows, np.int64, -1)
    m, mc, ms = _pad_lists(bm_rows, np.int64, -1)
    pos = np.full((b, top_k), -1, np.int32)
    row = np.full((b, top_k), -1, np.int64)
    score = np.zeros((b, top_k), np.float64)
    cnt = np.zeros(b, np.int32)
    cp, cn = (None, 0)
    if canon is not None:
        canon = np.ascontiguousarray(canon, np.uint32)
        cp, cn = canon.ctypes.data, canon.shape[0]
    check(lib().rag_fuse_rrf(v.ctypes.data, vc.ctypes.data, v.shape[1] if vs else 0, m.ctypes.data, mc.ctypes.data,
                             m.shape[1] if ms else 0, cp, cn, b, int(top_k), float(k_const), pos.ctypes.data,
                             row.ctypes.data, score.ctypes.data, cnt.ctypes.data), 'rag_fuse_rrf')
    return pos, row, score, cnt


def fuse_weighted(vec_rows, vec_scores, bm_rows, bm_scores, top_k, vector_weight=0.5, bm25_weight=0.5, canon=None):
    b = len(vec_rows)
    v, vc, vs = _pad_lists(vec_rows, np.int64, -1)
    vsc, _, _ = _pad_lists(vec_scores, np.float64, 0)
    m, mc, ms = _pad_lists(bm_rows, np.int64, -1)
    msc, _, _ = _pad_lists(bm_scores, np.float64, 0)
    pos = np.full((b, top_k), -1, np.int32)
    row = np.full((b, top_k), -1, np.int64)
    score = np.zeros((b, top_k), np.float64)
    cnt = np.zeros(b, np.int32)
    cp, cn = (None, 0)
    if canon is not None:
        canon = np.ascontiguousarray(canon, np.uint32)
        cp, cn = canon.ctypes.data, canon.shape[0]
    check(lib().rag_fuse_weighted(v.ctypes.data, vsc.ctypes.data, vc.ctypes.data, v.shape[1] if vs else 0,
                                  m.ctypes.data, msc.ctypes.data, mc.ctypes.data, m.shape[1] if ms else 0, cp, cn, b,
                                  int(top_k), float(vector_weight), float(bm25_weight), pos.ctypes.data,
                                  row.ctypes.data, score.ctypes.data, cnt.ctypes.data), 'rag_fuse_weighted')
    return pos, row, score, cnt


def _check_lists(vec_row, vec_cnt, bm_row, bm_cnt):
    b = vec_row.shape[0]
    if vec_row.dim() != 2 or bm_row.dim() != 2 or bm_row.shape[0] != b or tuple(vec_cnt.shape) != (b,) \
            or tuple(bm_cnt.shape) != (b,):
        raise ValueError("fusion inputs must be [B, stride] row tensors with [B] counts")


def fuse_rrf_dev(vec_row, vec_cnt, bm_row, bm_cnt, top_k, canon=None, k_const=60.0):
    import torch
    b = vec_row.shape[0]
    dev = vec_row.device
    pos = torch.empty((b, top_k), dtype=torch.int32, device=dev)
    row = torch.empty((b, top_k), dtype=torch.int64, device=dev)
    score = torch.empty((b, top_k), dtype=torch.float64, device=dev)
    cnt = torch.empty((b,), dtype=torch.int32, device=dev)
    _check_lists(vec_row, vec_cnt, bm_row, bm_cnt)
    check(lib().rag_fuse_rrf_dev(_ptr(vec_row, torch.int64), _ptr(vec_cnt, torch.int32), vec_row.shape[1],
                                 _ptr(bm_row, torch.int64), _ptr(bm_cnt, torch.int32), bm_row.shape[1],
                                 _ptr(canon, torch.uint32), 0 if canon is None else canon.shape[0], b, int(top_k),
                                 float(k_const), _ptr(pos), _ptr(row), _ptr(score), _ptr(cnt), _stream()),
          'rag_fuse_rrf_dev')
    return pos, row, score, cnt


def fuse_weighted_dev(vec_row, vec_score, vec_cnt, bm_row, bm_score, bm_cnt, top_k, vector_weight=0.5,
                      bm25_weight=0.5, canon=None):
    import torch
    b = vec_row.shape[0]
    dev = vec_row.device
    pos = torch.empty((b, top_k), dtype=torch.int32, device=dev)
    row = torch.empty((b, top_k), dtype=torch.int64, device=dev)
    score = torch.empty((b, top_k), dtype=torch.float64, device=dev)
    cnt = torch.empty((b,), dtype=torch.int32, device=dev)
    _check_lists(vec_row, vec_cnt, bm_row, bm_cnt)
    if vec_score.shape != vec_row.shape or bm_score.shape != bm_row.shape:
        raise ValueError("score tensors must have the shape of their row tensors")
    check(lib().rag_fuse_weighted_dev(_ptr(vec_row, torch.int64), _ptr(vec_score, torch.float64),
                                      _ptr(vec_cnt, torch.int32), vec_row.shape[1], _ptr(bm_row, torch.int64),
                                      _ptr(bm_score, torch.float64), _ptr(bm_cnt, torch.int32), bm_row.shape[1],
                                      _ptr(canon, torch.uint32), 0 if canon is None else canon.shape[0], b, int(top_k),
                                      float(vector_weight), float(bm25_weight), _ptr(pos), _ptr(row), _ptr(score),
                                      _ptr(cnt), _stream()), 'rag_fuse_weighted_dev')
    return pos, row, score, cnt


def topk_segmented(scores, seg_off, top_k):
    s, _ = _np(scores, np.float32)
    o, _ = _np(seg_off, np.int32)
    b = o.shape[0] - 1
    idx = np.full((b, top_k), -1, np.int32)
    cnt = np.zeros(b, np.int32)
    check(lib().rag_topk_segmented(s.ctypes.data, o.ctypes.data, b, int(top_k), idx.ctypes.data, cnt.ctypes.data),
          'rag_topk_segmented')
    return idx, cnt


def topk_segmented_dev(scores, seg_off, top_k):
    import torch
    b = seg_off.shape[0] - 1
    idx = torch.empty((b, top_k), dtype=torch.int32, device=scores.device)
    cnt = torch.empty((b,), dtype=torch.int32, device=scores.device)
    check(lib().rag_topk_segmented_dev(_ptr(scores, torch.float32), scores.numel(), _ptr(seg_off, torch.int32), b,
                                       int(top_k), _ptr(idx), _ptr(cnt), _stream()), 'rag_topk_segmented_dev')
    return idx, cnt


def merge_topk_dev(score, row, count, k_out, smaller_is_better=False):
    """[B, L, k_in] lists (fp32 or fp64 scores, int64 global rows) -> sorted [B, k_out]."""
    import torch
    b, nl, kin = score.shape
    dev = score.device
    if tuple(row.shape) != (b, nl, kin) or (count is not None and tuple(count.shape) != (b, nl)):
        raise ValueError("merge inputs must be [B, L, k_in] scores and rows with [B, L] counts")
    out_s = torch.empty((b, k_out), dtype=score.dtype, device=dev)
    out_r = torch.empty((b, k_out), dtype=torch.int64, device=dev)
    out_c = torch.empty((b,), dtype=torch.int32, device=dev)
    if score.dtype == torch.float32:
        check(lib().rag_merge_topk_f32_dev(_ptr(score), _ptr(row, torch.int64), _ptr(count, torch.int32), b, nl, kin,
                                           int(k_out),
                                           1 if smaller_is_better else 0, _ptr(out_s), _ptr(out_r), _ptr(out_c),
                                           _stream()), 'rag_merge_topk_f32_dev')
    elif score.dtype == torch.float64:
        check(lib().rag_merge_topk_f64_dev(_ptr(score), _ptr(row, torch.int64), _ptr(count, torch.int32), b, nl, kin,
                                           int(k_out), _ptr(out_s), _ptr(out_r), _ptr(out_c), _stream()),
              'rag_merge_topk_f64_dev')
    else:
        raise ValueError("scores must be float32 or float64")
    return out_s, out_r, out_c


def topk_stable_f64(scores, top_k):
    """Positions of the best ``top_k`` fp64 scores, best first, equal scores in input order
    (rag_merge_topk_f64_dev; its tie rule prefers the larger row, so position i travels as
    row n-1-i)."""
    import torch
    n = len(scores)
    if n == 0:
        return []
    dev = torch.device('cuda')
    s64 = torch.tensor(list(scores), dtype=torch.float64, device=dev).reshape(1, 1, n)
    rows = torch.arange(n - 1, -1, -1, dtype=torch.int64, device=dev).reshape(1, 1, n)
    _, r, c = merge_topk_dev(s64, rows, None, min(int(top_k), MAX_K))
    return [n - 1 - int(x) for x in r[0, :int(c[0])].cpu()]


def ce_pair_len_dev(query_off, chunk_off, chunk_row_base, pair_query, pair_row, max_len):
    import torch
    n = pair_row.shape[0]
    if tuple(pair_query.shape) != (n,):
        raise ValueError("pair_query and pair_row must have one entry per pair")
    out = torch.empty((n,), dtype=torch.int32, device=pair_row.device)
    check(lib().rag_ce_pair_len_dev(_ptr(query_off, torch.int64), query_off.shape[0] - 1, _ptr(chunk_off, torch.int64),
                                    chunk_off.shape[0] - 1, int(chunk_row_base), _ptr(pair_query, torch.int32),
                                    _ptr(pair_row, torch.int64), n, int(max_len), _ptr(out), _stream()),
          'rag_ce_pair_len_dev')
    return out


def ce_pack_dev(query_tok, query_off, chunk_tok, chunk_off, chunk_row_base, pair_query, pair_row, pair_slot,
                n_slots, seq_len, cls_id=101, sep_id=102, pad_id=0):
    import torch
    dev = pair_row.device
    ids = torch.empty((n_slots, seq_len), dtype=torch.int64, device=dev)
    types = torch.empty((n_slots, seq_len), dtype=torch.int64, device=dev)
    mask = torch.empty((n_slots, seq_len), dtype=torch.int64, device=dev)
    n = pair_row.shape[0]
    if tuple(pair_query.shape) != (n,) or (pair_slot is not None and tuple(pair_slot.shape) != (n,)):
        raise ValueError("pair_query, pair_row and pair_slot must have one entry per pair")
    check(lib().rag_ce_pack_dev(_ptr(query_tok, torch.int32), _ptr(query_off, torch.int64), query_off.shape[0] - 1,
                                _ptr(chunk_tok, torch.int32), _ptr(chunk_off, torch.int64), chunk_off.shape[0] - 1,
                                int(chunk_row_base), _ptr(pair_query, torch.int32), _ptr(pair_row, torch.int64),
                                _ptr(pair_slot, torch.int32), n, int(n_slots), int(seq_len), int(cls_id), int(sep_id),
                                int(pad_id), _ptr(ids), _ptr(types), _ptr(mask), _stream()), 'rag_ce_pack_dev')
    return ids, types, mask


# ---- transformer encoder pieces ----------------------------------------------------------------
def enc_attention_supported(dtype, head_dim, seq_len):
    import torch
    return dtype in (torch.bfloat16, torch.float16, torch.float32) and head_dim in (32, 64) and seq_len <= ENC_MAX_SEQ


def _enc_dtype(t):
    import torch
    if t.dtype == torch.bfloat16:
        return DTYPE_BF16
    if t.dtype == torch.float16:
        return DTYPE_F16
    if t.dtype == torch.float32:
        return DTYPE_F32
    raise ValueError(f"the encoder kernels handle bfloat16, float16 and float32, not {t.dtype}")


def _is_16bit(dtype):
    import torch
    return dtype in (torch.bfloat16, torch.float16)


def enc_attention_dev(qkv, lens, n_heads):
    """qkv [B, S, 3*H] bf16 (query | key | value, heads inside each), lens [B] int32 ->
    context [B, S, H] bf16 (rag_enc_attention_dev)."""
    import torch
    b, s, h3 = qkv.shape
    hidden = h3 // 3
    out = torch.empty((b, s, hidden), dtype=qkv.dtype, device=qkv.device)
    if tuple(lens.shape) != (b,):
        raise ValueError("lens must hold one length per sequence")
    check(lib().rag_enc_attention_dev(_ptr(qkv), _ptr(lens, torch.int32), b, s, int(n_heads),
                                      hidden // int(n_heads), _enc_dtype(qkv), _ptr(out), _stream()),
          'rag_enc_attention_dev')
    return out


def enc_add_layernorm_dev(x, residual, gamma, beta, eps):
    """LayerNorm(x + residual) over the last dimension, one kernel (rag_enc_add_layernorm_dev)."""
    import torch
    if x.dtype not in (torch.bfloat16, torch.float16, torch.float32):
        raise ValueError("enc_add_layernorm_dev handles bfloat16, float16 and float32")
    hidden = x.shape[-1]
    rows = x.numel() // hidden
    if residual is not None and residual.shape != x.shape:
        raise ValueError("residual must have the shape of x")
    if tuple(gamma.shape) != (hidden,) or tuple(beta.shape) != (hidden,):
        raise ValueError("gamma and beta must have one entry per hidden unit")
    out = torch.empty_like(x)
    check(lib().rag_enc_add_layernorm_dev(_ptr(x), _ptr(residual, x.dtype), _ptr(gamma, x.dtype), _ptr(beta, x.dtype),
                                          float(eps), rows, hidden, _enc_dtype(x),
                                          _ptr(out), _stream()), 'rag_enc_add_layernorm_dev')
    return out


def enc_attention_packed_dev(qkv, cu_seqlens, max_seq_len, n_heads):
    """Packed batch: qkv [T, 3*H] bf16, cu_seqlens [n_seq + 1] int32 -> context [T, H]
    (rag_enc_attention_packed_dev)."""
    import torch
    t, h3 = qkv.shape
    hidden = h3 // 3
    out = torch.empty((t, hidden), dtype=qkv.dtype, device=qkv.device)
    n_seq = cu_seqlens.shape[0] - 1
    check(lib().rag_enc_attention_packed_dev(_ptr(qkv), _ptr(cu_seqlens, torch.int32), n_seq, t,
                                             int(max_seq_len), int(n_heads), hidden // int(n_heads), _enc_dtype(qkv),
                                             _ptr(out), _stream()), 'rag_enc_attention_packed_dev')
    return out


def enc_embed_layernorm_dev(ids, types, pos, word, position, type_emb, gamma, beta, eps):
    """LayerNorm(word[ids] + position[pos] + type_emb[types]) for a flat token list
    (rag_enc_embed_layernorm_dev); ids/types/pos int32 [T]."""
    import torch
    t = ids.shape[0]
    hidden = word.shape[1]
    dt = word.dtype
    if dt not in (torch.bfloat16, torch.float16, torch.float32):
        raise ValueError("enc_embed_layernorm_dev handles bfloat16, float16 and float32 tables")
    if tuple(pos.shape) != (t,) or (types is not None and tuple(types.shape) != (t,)):
        raise ValueError("ids, types and pos must have one entry per token")
    out = torch.empty((t, hidden), dtype=dt, device=ids.device)
    check(lib().rag_enc_embed_layernorm_dev(_ptr(ids, torch.int32), _ptr(types, torch.int32), _ptr(pos, torch.int32),
                                            _ptr(word), _ptr(position, dt), _ptr(type_emb, dt), _ptr(gamma, dt),
                                            _ptr(beta, dt), float(eps), t, hidden, word.shape[0], position.shape[0],
                                            type_emb.shape[0], _enc_dtype(word),
                                            _ptr(out), _stream()), 'rag_enc_embed_layernorm_dev')
    return out


def enc_embed_layernorm_split_dev(ids, types, pos, word, position, type_emb, gamma, beta, eps, half=None, want_out=True):
    """enc_embed_layernorm_dev for float32 tables that also returns the rows split for the first split-precision GEMM
    (rag_enc_embed_layernorm_split_dev): (out fp32 [T, H] or None, [T, 3H] = [hi | hi | lo] in `half`)."""
    import torch
    t = ids.shape[0]
    hidden = word.shape[1]
    if word.dtype != torch.float32:
        raise ValueError("enc_embed_layernorm_split_dev handles float32 tables")
    if tuple(pos.shape) != (t,) or (types is not None and tuple(types.shape) != (t,)):
        raise ValueError("ids, types and pos must have one entry per token")
    code, hdt = _half_code(half)
    out = torch.empty((t, hidden), dtype=torch.float32, device=ids.device) if want_out else None
    split = torch.empty((t, 3 * hidden), dtype=hdt, device=ids.device)
    f32 = torch.float32
    check(lib().rag_enc_embed_layernorm_split_dev(_ptr(ids, torch.int32), _ptr(types, torch.int32), _ptr(pos, torch.int32),
                                                  _ptr(word, f32), _ptr(position, f32), _ptr(type_emb, f32), _ptr(gamma, f32),
                                                  _ptr(beta, f32), float(eps), t, hidden, word.shape[0], position.shape[0],
                                                  type_emb.shape[0], _ptr(out), code, _ptr(split), _stream()),
          'rag_enc_embed_layernorm_split_dev')
    return out, split


def ce_pack_packed_dev(query_tok, query_off, chunk_tok, chunk_off, chunk_row_base, pair_query, pair_row, cu,
                       total_tokens, max_len, cls_id=101, sep_id=102, padded_total=None):
    """Unpadded token stream of the pairs: (ids, types, positions) int32 [total_tokens]
    (rag_ce_pack_packed_dev); cu [n_pairs + 1] int32 = running sum of ce_pair_len_dev."""
    import torch
    n = pair_row.shape[0]
    if tuple(pair_query.shape) != (n,) or tuple(cu.shape) != (n + 1,):
        raise ValueError("pair_query / pair_row need one entry per pair and cu one more")
    dev = pair_row.device
    # outputs may carry unused tail tokens (id 0, position 0) up to `padded_total`
    size = int(total_tokens if padded_total is None else padded_total)
    if size < total_tokens:
        raise ValueError("padded_total is smaller than total_tokens")
    ids = torch.zeros((size,), dtype=torch.int32, device=dev)
    types = torch.zeros((size,), dtype=torch.int32, device=dev)
    pos = torch.zeros((size,), dtype=torch.int32, device=dev)
    check(lib().rag_ce_pack_packed_dev(_ptr(query_tok, torch.int32), _ptr(query_off, torch.int64), query_off.shape[0] - 1,
                                       _ptr(chunk_tok, torch.int32), _ptr(chunk_off, torch.int64), chunk_off.shape[0] - 1,
                                       int(chunk_row_base), _ptr(pair_query, torch.int32), _ptr(pair_row, torch.int64),
                                       _ptr(cu, torch.int32), n, int(total_tokens), int(max_len), int(cls_id), int(sep_id),
                                       _ptr(ids), _ptr(types), _ptr(pos), _stream()), 'rag_ce_pack_packed_dev')
    return ids, types, pos


def enc_mean_pool_dev(x, cu_seqlens, normalize=True):
    """Per-sequence mean of a packed fp32 batch x [T, H], optionally L2-normalised -> [n_seq, H]
    (rag_enc_mean_pool_dev)."""
    import torch
    t, hidden = x.shape
    n_seq = cu_seqlens.shape[0] - 1
    out = torch.empty((n_seq, hidden), dtype=torch.float32, device=x.device)
    check(lib().rag_enc_mean_pool_dev(_ptr(x, torch.float32), _ptr(cu_seqlens, torch.int32), n_seq, t, hidden,
                                      1 if normalize else 0, _ptr(out), _stream()), 'rag_enc_mean_pool_dev')
    return out


def enc_split_bf16x3_dev(x, half=None):
    """fp32 [rows, K] -> `half` (bfloat16 unless given) [rows, 3K] = [hi | hi | lo] (rag_enc_split_bf16x3_dev)."""
    import torch
    rows, k = x.shape
    code, hdt = _half_code(half)
    out = torch.empty((rows, 3 * k), dtype=hdt, device=x.device)
    check(lib().rag_enc_split_bf16x3_dev(_ptr(x, torch.float32), rows, k, code, _ptr(out), _stream()),
          'rag_enc_split_bf16x3_dev')
    return out


def enc_bias_act_split_dev(y, bias, gelu, half=None):
    """fp32 y [rows, K], bias [K] -> bf16 [rows, 3K] = split of act(y + bias), act = exact GELU or identity
    (rag_enc_bias_act_split_bf16x3_dev)."""
    import torch
    rows, k = y.shape
    if tuple(bias.shape) != (k,):
        raise ValueError("bias must have one entry per column")
    code, hdt = _half_code(half)
    out = torch.empty((rows, 3 * k), dtype=hdt, device=y.device)
    check(lib().rag_enc_bias_act_split_bf16x3_dev(_ptr(y, torch.float32), _ptr(bias, torch.float32), 1 if gelu else 0,
                                                  rows, k, code, _ptr(out), _stream()), 'rag_enc_bias_act_split_bf16x3_dev')
    return out


def enc_bias_add_layernorm_split_dev(y, bias, residual, gamma, beta, eps, want_split=True, half=None):
    """fp32 LayerNorm((y + bias) + residual) -> (out fp32 [rows, H], its bf16 split [rows, 3H] or None)
    (rag_enc_bias_add_layernorm_split_dev)."""
    import torch
    rows, h = y.shape
    if residual.shape != y.shape:
        raise ValueError("residual must have the shape of y")
    if tuple(bias.shape) != (h,) or tuple(gamma.shape) != (h,) or tuple(beta.shape) != (h,):
        raise ValueError("bias, gamma and beta must have one entry per hidden unit")
    out = torch.empty_like(y)
    code, hdt = _half_code(half)
    split = torch.empty((rows, 3 * h), dtype=hdt, device=y.device) if want_split else None
    check(lib().rag_enc_bias_add_layernorm_split_dev(_ptr(y, torch.float32), _ptr(bias, torch.float32),
                                                     _ptr(residual, torch.float32), _ptr(gamma, torch.float32),
                                                     _ptr(beta, torch.float32), float(eps), rows, h, code, _ptr(out),
                                                     _ptr(split) if split is not None else None, _stream()),
          'rag_enc_bias_add_layernorm_split_dev')
    return out, split


def split_weight_bf16x3(w, half=None):
    """fp32 weight [N, K] -> `half` (bfloat16 unless given) [N, 3K] = [w_hi | w_lo | w_hi], the partner of
    enc_split_bf16x3_dev."""
    import torch
    _, hdt = _half_code(half)
    hi = w.to(hdt)
    lo = (w - hi.float()).to(hdt)
    return torch.cat([hi, lo, hi], dim=1).contiguous()


def enc_attention_packed_first_dev(qkv, cu_seqlens, max_seq_len, n_heads):
    """Attention output of the first token of every sequence only: qkv [T, 3*H] bf16 or fp16 ->
    [n_seq, H] (rag_enc_attention_packed_first_dev)."""
    import torch
    if not _is_16bit(qkv.dtype):
        raise ValueError("enc_attention_packed_first_dev handles bfloat16 and float16")
    t, h3 = qkv.shape
    hidden = h3 // 3
    n_seq = cu_seqlens.shape[0] - 1
    out = torch.empty((n_seq, hidden), dtype=qkv.dtype, device=qkv.device)
    check(lib().rag_enc_attention_packed_first_dev(_ptr(qkv), _ptr(cu_seqlens, torch.int32), n_seq, t,
                                                   int(max_seq_len), int(n_heads), hidden // int(n_heads), _enc_dtype(qkv),
                                                   _ptr(out), _stream()), 'rag_enc_attention_packed_first_dev')
    return out


def enc_attention_split_supported(head_dim, max_seq_len):
    """Whether rag_enc_attention_split_dev covers this head size and sequence length (LDS holds a
    head's keys and values twice: bf16 high and low parts)."""
    if head_dim not in (32, 64) or max_seq_len > ENC_MAX_SEQ:
        return False
    kpad = (int(max_seq_len) + 31) // 32 * 32
    if head_dim == 32:
        return kpad * 256 <= 160 * 1024
    return 2 * (kpad * (head_dim + 8) * 2 + head_dim * (kpad + 4) * 2) <= 160 * 1024


def _half_code(half):
    """RAG_DTYPE_* code of the 16-bit type split-precision operands are held in (None = bfloat16)."""
    import torch
    if half is None or half == torch.bfloat16:
        return DTYPE_BF16, torch.bfloat16
    if half == torch.float16:
        return DTYPE_F16, torch.float16
    raise ValueError(f"split-precision halves are bfloat16 or float16, not {half}")


def enc_attention_split_dev(qkv, bias, cu_seqlens, max_seq_len, n_heads, first_only=False, half=None, hi_lo=False):
    """fp32-equivalent attention of a packed batch (rag_enc_attention_split_dev): qkv [T, 3*H] fp32 without
    its bias, bias [3*H] fp32 or None -> `half` [T or n_seq, 3*H] = [hi | hi | lo] of the attention output, or with
    hi_lo (rag_enc_attention_split_hl_dev) [T or n_seq, 2*H] = [hi | lo]."""
    import torch
    code, hdt = _half_code(half)
    t, h3 = qkv.shape
    n_seq = cu_seqlens.shape[0] - 1
    if bias is not None and tuple(bias.shape) != (h3,):
        raise ValueError("bias must have one entry per projection column")
    out = torch.empty((n_seq if first_only else t, 2 * (h3 // 3) if hi_lo else h3), dtype=hdt, device=qkv.device)
    fn = lib().rag_enc_attention_split_hl_dev if hi_lo else lib().rag_enc_attention_split_dev
    check(fn(_ptr(qkv, torch.float32), _ptr(bias, torch.float32), _ptr(cu_seqlens, torch.int32), n_seq, t, int(max_seq_len),
             int(n_heads), (h3 // 3) // int(n_heads), 1 if first_only else 0, code, _ptr(out), _stream()),
          'rag_enc_attention_split_hl_dev' if hi_lo else 'rag_enc_attention_split_dev')
    return out


def enc_gelu_(x):
    """In-place exact GELU (rag_enc_gelu_dev); returns x."""
    check(lib().rag_enc_gelu_dev(_ptr(x), x.numel(), _enc_dtype(x), _stream()), 'rag_enc_gelu_dev')
    return x




def enc_mlp_supported(x, w1, w2):
    import torch
    return (_is_16bit(x.dtype) and w1.dtype == x.dtype and w2.dtype == x.dtype
            and x.shape[-1] == 384 and w1.shape[1] == 384 and w1.shape[0] % 32 == 0)


def enc_mlp_dev(x, w1, b1, w2, b2, gamma, beta, eps=1e-12):
    """LayerNorm(x + GELU(x @ w1.T + b1) @ w2.T + b2) * gamma + beta in one kernel
    (rag_enc_mlp_dev); x [M, 384] bf16 or fp16 (every operand of that type), w1 [I, 384], w2 [384, I]."""
    import torch
    m, hid = x.shape
    inter = w1.shape[0]
    if tuple(w1.shape) != (inter, hid) or tuple(w2.shape) != (hid, inter) or tuple(b1.shape) != (inter,):
        raise ValueError("w1 must be [I, H], w2 [H, I] and b1 [I]")
    if tuple(b2.shape) != (hid,) or tuple(gamma.shape) != (hid,) or tuple(beta.shape) != (hid,):
        raise ValueError("b2, gamma and beta must be [H]")
    y = torch.empty_like(x)
    bf = x.dtype
    if not _is_16bit(bf):
        raise ValueError("enc_mlp_dev handles bfloat16 and float16")
    check(lib().rag_enc_mlp_dev(_ptr(x, bf), _ptr(w1, bf), _ptr(b1, bf), _ptr(w2, bf), _ptr(b2, bf), _ptr(gamma, bf),
                                _ptr(beta, bf), float(eps), m, hid, inter, _enc_dtype(x), _ptr(y), _stream()),
          'rag_enc_mlp_dev')
    return y


def enc_linear384_supported(x, weight):
    import torch
    return (_is_16bit(x.dtype) and weight.dtype == x.dtype and x.shape[-1] == 384
            and weight.shape[1] == 384 and weight.shape[0] % 32 == 0 and weight.shape[0] <= 6144)


def enc_linear384_dev(x, weight, bias, residual=None, gamma=None, beta=None, eps=1e-12):
    """x @ weight.T + bias for 384-wide bf16 or fp16 inputs (rag_enc_linear384_dev); x [M, 384], weight [N, 384].
    With ``residual`` [M, 384] (N == 384): LayerNorm(that + residual) * gamma + beta."""
    import torch
    m, k = x.shape
    n = weight.shape[0]
    if k != 384 or weight.shape[1] != 384 or tuple(bias.shape) != (n,):
        raise ValueError("x must be [M, 384], weight [N, 384] and bias [N]")
    if residual is not None and (n != 384 or tuple(residual.shape) != (m, 384) or gamma is None or beta is None
                                 or tuple(gamma.shape) != (384,) or tuple(beta.shape) != (384,)):
        raise ValueError("the LayerNorm tail needs N == 384, residual [M, 384], gamma [384] and beta [384]")
    bf = x.dtype
    if not _is_16bit(bf):
        raise ValueError("enc_linear384_dev handles bfloat16 and float16")
    y = torch.empty((m, n), dtype=bf, device=x.device)
    check(lib().rag_enc_linear384_dev(_ptr(x, bf), _ptr(weight, bf), _ptr(bias, bf), m, n, _ptr(residual, bf),
                                      _ptr(gamma, bf), _ptr(beta, bf), float(eps), _enc_dtype(x), _ptr(y), _stream()),
          'rag_enc_linear384_dev')
    return y


QKV_ATTENTION_TILE = 256     # tokens of a tile of enc_qkv_attention_dev, and its longest sequence


def enc_qkv_attention_tiles(cu_host):
    """Cuts a packed batch into the tiles of enc_qkv_attention_dev (rag_enc_qkv_attention_tiles): cu_host = token offsets
    of the sequences on the HOST (int64, n_seq + 1 entries) -> int32 array [n_tiles + 1], entry t = first sequence of
    tile t; whole sequences, at most 256 tokens a tile."""
    cu = np.ascontiguousarray(cu_host, dtype=np.int64)
    n_seq = len(cu) - 1
    out = np.zeros(max(n_seq, 0) + 1, np.int32)
    n = C.c_int32(0)
    check(load_library().rag_enc_qkv_attention_tiles(cu.ctypes.data, n_seq, out.ctypes.data, C.addressof(n)),   # host only
          'rag_enc_qkv_attention_tiles')
    return out[:n.value + 1].copy()


def enc_qkv_attention_pack(lens_host):
    """An order of the sequences (lens_host: tokens of each, on the HOST) in which consecutive ones fill the 256-token
    tiles of enc_qkv_attention_dev well (rag_enc_qkv_attention_pack) -> (order int64 [n], tile_seq int32 [n_tiles + 1]:
    position in that order of every tile's first sequence); empty sequences come last, in no tile.  None if a sequence
    is longer than a tile."""
    lens = np.ascontiguousarray(lens_host, dtype=np.int32)
    if lens.size and int(lens.max()) > QKV_ATTENTION_TILE:
        return None
    order = np.zeros(lens.size, np.int64)
    tiles = np.zeros(lens.size + 1, np.int32)
    n = C.c_int32(0)
    check(load_library().rag_enc_qkv_attention_pack(lens.ctypes.data, lens.size, order.ctypes.data, tiles.ctypes.data, C.addressof(n)),
          'rag_enc_qkv_attention_pack')
    return order, tiles[:n.value + 1].copy()


def enc_qkv_attention_dev(x, weight, bias, cu_seqlens, tile_seq, n_heads, first_only=False):
    """Self-attention output of a packed batch straight from the layer input: x [T, 384] bf16 or fp16, weight [1152, 384]
    and bias [1152] of the q | k | v projection, cu_seqlens [n_seq + 1] int32 and tile_seq (enc_qkv_attention_tiles, int32)
    on the device -> context [T, 384] (rag_enc_qkv_attention_dev); first_only: the first token's row of every sequence,
    [n_seq, 384] (rag_enc_qkv_attention_first_dev)."""
    import torch
    t, k = x.shape
    if k != 384 or tuple(weight.shape) != (1152, 384) or tuple(bias.shape) != (1152,):
        raise ValueError("x must be [T, 384], weight [1152, 384] and bias [1152]")
    dt = x.dtype
    if not _is_16bit(dt):
        raise ValueError("enc_qkv_attention_dev handles bfloat16 and float16")
    n_seq, n_tiles = cu_seqlens.shape[0] - 1, tile_seq.shape[0] - 1
    out = torch.empty((n_seq if first_only else t, 384), dtype=dt, device=x.device)
    fn = lib().rag_enc_qkv_attention_first_dev if first_only else lib().rag_enc_qkv_attention_dev
    check(fn(_ptr(x, dt), _ptr(weight, dt), _ptr(bias, dt), _ptr(cu_seqlens, torch.int32), n_seq, t,
             _ptr(tile_seq, torch.int32), n_tiles, n_heads, 384 // n_heads, _enc_dtype(x), _ptr(out), _stream()),
          'rag_enc_qkv_attention_first_dev' if first_only else 'rag_enc_qkv_attention_dev')
    return out


def enc_mlp_profile(enable=True):
    """Measurement aid: record a HIP event pair around every feed-forward kernel launch."""
    check(lib().rag_enc_mlp_profile(1 if enable else 0), 'rag_enc_mlp_profile')


def enc_mlp_profile_read():
    """(durations in ms, row counts) of the feed-forward kernel launches since the last read."""
    ms = np.zeros(1024, np.float32)
    rows = np.zeros(1024, np.int64)
    n = _i32(0)
    check(lib().rag_enc_mlp_profile_read(ms.ctypes.data, rows.ctypes.data, 1024, C.byref(n)), 'rag_enc_mlp_profile_read')
    return ms[:n.value].copy(), rows[:n.value].copy()


def enc_attn_out_mlp_dev(ctx, x, wo, bo, gamma1, beta1, w1, b1, w2, b2, gamma2, beta2, eps=1e-12):
    """Everything of a BERT layer after self-attention in one kernel (rag_enc_attn_out_mlp_dev):
    x1 = LN(x + ctx @ wo.T + bo), y = LN(x1 + GELU(x1 @ w1.T + b1) @ w2.T + b2); ctx, x [M, 384] bf16."""
    import torch
    m, hid = x.shape
    inter = w1.shape[0]
    if tuple(ctx.shape) != (m, hid) or tuple(wo.shape) != (hid, hid):
        raise ValueError("ctx must be [M, H] and wo [H, H]")
    if tuple(w1.shape) != (inter, hid) or tuple(w2.shape) != (hid, inter) or tuple(b1.shape) != (inter,):
        raise ValueError("w1 must be [I, H], w2 [H, I] and b1 [I]")
    for v in (bo, gamma1, beta1, b2, gamma2, beta2):
        if tuple(v.shape) != (hid,):
            raise ValueError("bo, b2 and the LayerNorm vectors must be [H]")
    y = torch.empty_like(x)
    bf = x.dtype
    if not _is_16bit(bf):
        raise ValueError("enc_attn_out_mlp_dev handles bfloat16 and float16")
    check(lib().rag_enc_attn_out_mlp_dev(_ptr(ctx, bf), _ptr(x, bf), _ptr(wo, bf), _ptr(bo, bf), _ptr(gamma1, bf),
                                         _ptr(beta1, bf), _ptr(w1, bf), _ptr(b1, bf), _ptr(w2, bf), _ptr(b2, bf),
                                         _ptr(gamma2, bf), _ptr(beta2, bf), float(eps), m, hid, inter, _enc_dtype(x),
                                         _ptr(y), _stream()), 'rag_enc_attn_out_mlp_dev')
    return y


def enc_norm_mlp_dev(proj, x, gamma1, beta1, w1, b1, w2, b2, gamma2, beta2, eps=1e-12):
    """A BERT layer after its attention-output projection in one kernel (rag_enc_norm_mlp_dev):
    x1 = LN(x + proj), y = LN(x1 + GELU(x1 @ w1.T + b1) @ w2.T + b2); proj, x [M, 384] bf16."""
    import torch
    m, hid = x.shape
    inter = w1.shape[0]
    if tuple(proj.shape) != (m, hid):
        raise ValueError("proj must have the shape of x")
    if tuple(w1.shape) != (inter, hid) or tuple(w2.shape) != (hid, inter) or tuple(b1.shape) != (inter,):
        raise ValueError("w1 must be [I, H], w2 [H, I] and b1 [I]")
    for v in (gamma1, beta1, b2, gamma2, beta2):
        if tuple(v.shape) != (hid,):
            raise ValueError("b2 and the LayerNorm vectors must be [H]")
    y = torch.empty_like(x)
    bf = x.dtype
    if not _is_16bit(bf):
        raise ValueError("enc_norm_mlp_dev handles bfloat16 and float16")
    check(lib().rag_enc_norm_mlp_dev(_ptr(proj, bf), _ptr(x, bf), _ptr(gamma1, bf), _ptr(beta1, bf), _ptr(w1, bf),
                                     _ptr(b1, bf), _ptr(w2, bf), _ptr(b2, bf), _ptr(gamma2, bf), _ptr(beta2, bf),
                                     float(eps), m, hid, inter, _enc_dtype(x), _ptr(y), _stream()), 'rag_enc_norm_mlp_dev')
    return y


def split_weight_halves(w):
    """fp32 weight [384, K] on the GPU -> the weight operands of rag_enc_linear_ln_split_dev: the fp16 halves
    (hi = fp16(w), lo = fp16(w - hi)), each packed by 32-column chunk in the order the kernel streams them
    (rag_enc_linear_ln_split_pack_dev).  The tensors keep the shape [384, K] but not that layout:
    unpack_weight_halves gives the plain matrices back."""
    import torch
    n, k = w.shape
    w = w.contiguous().float()
    wh = torch.empty((n, k), dtype=torch.float16, device=w.device)
    wl = torch.empty((n, k), dtype=torch.float16, device=w.device)
    check(lib().rag_enc_linear_ln_split_pack_dev(_ptr(w, torch.float32), k, n, _ptr(wh), _ptr(wl), _stream()),
          'rag_enc_linear_ln_split_pack_dev')
    return wh, wl


def unpack_weight_halves(packed):
    """A packed image of split_weight_halves -> the plain [384, K] matrix (tests, inspection)."""
    import torch
    n, k = packed.shape
    f = torch.arange(n, device=packed.device)
    fo = (f & ~12) | ((f & 4) << 1) | ((f & 8) >> 1)
    pos = torch.arange(4, device=packed.device)
    j = pos[None, :] ^ ((f[:, None] >> 2) & 3)                      # image row f, position pos holds piece j
    img = packed.reshape(k // 32, n, 4, 8)
    out = torch.empty((k // 32, n, 4, 8), dtype=packed.dtype, device=packed.device)
    out[:, fo[:, None].expand(n, 4), j] = img[:, f[:, None].expand(n, 4), pos[None, :].expand(n, 4)]
    return out.permute(1, 0, 2, 3).reshape(n, k).contiguous()


def enc_linear_ln_split_supported(k, n, half):
    import torch
    return n == 384 and k % 32 == 0 and 32 <= k <= 2048 and half == torch.float16


def enc_linear_ln_split_dev(x_in, b1, wh, wl, b2, residual, gamma, beta, eps, want_split=True, want_y=True):
    """LayerNorm(residual + a @ W.T + b2) * gamma + beta at fp32 precision on the fp16 matrix cores
    (rag_enc_linear_ln_split_dev).  x_in fp32 [M, K] with b1: a = GELU(x_in + b1); x_in fp16 [M, 3K] = [hi | hi | lo]
    or [M, 2K] = [hi | lo] (b1 None): a is that split operand.  wh, wl: split_weight_halves(W), W [384, K].
    residual: fp32 [M, 384], or fp16 [M, 1152] = the split a previous call returned (rag_enc_linear_ln_split_chain_dev).
    Returns (y fp32 [M, 384] or None, its fp16 [hi | hi | lo] split [M, 1152] or None)."""
    import torch
    m = x_in.shape[0]
    n, k = wh.shape
    is_split = x_in.dtype == torch.float16
    if is_split and tuple(x_in.shape) == (m, 2 * k):
        is_split = 2
    elif tuple(x_in.shape) != (m, 3 * k if is_split else k) or (not is_split and x_in.dtype != torch.float32):
        raise ValueError("x_in must be fp32 [M, K] or fp16 [M, 3K] or [M, 2K]")
    if not is_split and (b1 is None or tuple(b1.shape) != (k,)):
        raise ValueError("an fp32 input needs its bias b1 [K]")
    chain = residual.dtype == torch.float16
    if tuple(wl.shape) != (n, k) or tuple(residual.shape) != (m, 3 * n if chain else n):
        raise ValueError("wl must have the shape of wh and residual must be fp32 [M, N] or fp16 [M, 3N]")
    if not chain and residual.dtype != torch.float32:
        raise ValueError("residual must be fp32 [M, N] or fp16 [M, 3N]")
    if not want_y and not want_split:
        raise ValueError("want_y=False needs want_split")
    for v in (b2, gamma, beta):
        if tuple(v.shape) != (n,):
            raise ValueError("b2, gamma and beta must be [N]")
    y = torch.empty((m, n), dtype=torch.float32, device=x_in.device) if want_y else None
    ys = torch.empty((m, 3 * n), dtype=torch.float16, device=x_in.device) if want_split else None
    f32 = torch.float32
    name = 'rag_enc_linear_ln_split_chain_dev' if chain else 'rag_enc_linear_ln_split_dev'
    check(getattr(lib(), name)(_ptr(x_in), int(is_split), None if is_split else _ptr(b1, f32),
                               _ptr(wh, torch.float16), _ptr(wl, torch.float16), _ptr(b2, f32),
                               _ptr(residual, torch.float16 if chain else f32), _ptr(gamma, f32), _ptr(beta, f32), float(eps),
                               m, k, n, _ptr(y) if y is not None else None, _ptr(ys) if ys is not None else None, _stream()),
          name)
    return y, ys


def enc_linear_ln_split_profile(enable=True):
    check(lib().rag_enc_linear_ln_split_profile(1 if enable else 0), 'rag_enc_linear_ln_split_profile')


def enc_linear_ln_split_profile_read():
    """(durations in ms, row counts, k) of the rag_enc_linear_ln_split_dev launches since the last read."""
    ms = np.zeros(1024, np.float32)
    rows = np.zeros(1024, np.int64)
    ks = np.zeros(1024, np.int32)
    n = _i32(0)
    check(lib().rag_enc_linear_ln_split_profile_read(ms.ctypes.data, rows.ctypes.data, ks.ctypes.data, 1024, C.byref(n)),
          'rag_enc_linear_ln_split_profile_read')
    return ms[:n.value].copy(), rows[:n.value].copy(), ks[:n.value].copy()
