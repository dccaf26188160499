// BM25 term-at-a-time scoring over term-major CSR postings with fp64 LDS accumulators and an
// exact top-K.  gfx950 (MI355X) only.
//
// Replaces rank_bm25.BM25Okapi.get_scores + np.argsort behind BM25Indexer.search (reference
// src/indexing/bm25_indexer.py:127-128).
//
// Layout in HBM: indptr[V+1] i64, rows[nnz] u32 (ascending inside a term), impact[nnz] f64,
// idf[V] f64.  Algorithmic traffic per query: 12 bytes per posting of each query token.
//
// Kernel 1 (bm25_offsets_kernel): for every query token and every document tile, the position
// of the tile's first posting inside the token's posting list (binary search).
// Kernel 2 (bm25_tile_kernel): one workgroup per (document tile, query), tiles outermost in the
// grid.  The tile's fp64 scores live in LDS; the query's tokens are applied one after the other
// (a barrier in between), so each document's sum is formed in query-token order exactly like the
// reference's `score += idf * impact` loop: bit-identical fp64 results.  The next token's postings are
// already in flight while a token is added.  The tile then emits its
// candidates for the query's top-K: documents whose score reaches the query's shared threshold,
// cut to the tile's K best; the K-th best of a tile raises the shared threshold (a lower bound
// of the final K-th score), which prunes later tiles.  Candidates go to one append-only list per
// query (a tile reserves its places with one atomic add).
// Kernel 3 (merge_topk_kernel): per query, the K best of the query's candidate list, sorted.
// With 64 tiles or more the tile kernel runs twice: first on a sample of one tile in sixteen, whose
// merged K-th score seeds the shared threshold, then on the rest, which then emit about 16 K
// candidates per query in all instead of dozens per tile.
#include "common.h"

#include <mutex>
#include "merge.h"

#include <stdlib.h>

namespace {

constexpr int BM_NT = 256;
constexpr int BM_TILE = 2048;      // documents per tile (16 KB of fp64 accumulators)
constexpr int BM_TOK = 32;         // query tokens whose posting ranges are looked up in one go
constexpr int BM_PF = 3;           // postings per thread fetched one token ahead (2, 3, 4 measured alike)
constexpr int BM_CTR_STRIDE = 32;  // ints between two queries' list counters (128 bytes)

struct Bm25Params {
    const int64_t* indptr;
    const uint32_t* rows;
    const double* impact;
    const double* idf;
    int64_t n_docs;
    int64_t vocab;
    uint32_t* fault;               // RAG_FAULT_* bits of this device (may be null)
    const int32_t* qterms;
    const int32_t* qoff;
    int n_queries;
    int n_terms_total;
    uint32_t* toff;                // [n_terms_total][n_tiles + 1]
    int n_tiles;
    int k;
    unsigned long long* tau;       // [n_queries] shared score-key threshold
    uint64_t* cand_key;            // [n_queries][n_tiles * k]: one append-only candidate list per query
    uint32_t* cand_row;
    int32_t* cand_count;           // [n_queries] candidates appended so far (read by the merge)
    int32_t* cand_ctr;             // the counters the tiles add to: one per query, BM_CTR_STRIDE ints apart (a cache
                                   // line each: thousands of tiles reserve places at the same time, and atomics on
                                   // one line are served one after the other)
    double* dense_out;             // scores mode: [n_docs]
    int share_tau;                 // tuning aid: 0 = tiles do not share their thresholds
    int tile_base;                 // first tile of this launch
};

__global__ void bm25_offsets_kernel(Bm25Params p) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = p.n_tiles + 1;
    if (gid >= (int64_t)p.n_terms_total * per) return;
    const int ti = (int)(gid / per);
    const int tile = (int)(gid - (int64_t)ti * per);
    const int term = p.qterms[ti];
    uint32_t result = 0;
    if (term >= p.vocab && tile == 0) rag_flag(p.fault, RAG_FAULT_BM25_TERM);
    if (term >= 0 && term < p.vocab) {
        const int64_t lo0 = p.indptr[term], hi0 = p.indptr[term + 1];
        const uint64_t target = (uint64_t)tile * BM_TILE;
        int64_t lo = lo0, hi = hi0;
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if ((uint64_t)p.rows[mid] < target) lo = mid + 1; else hi = mid;
        }
        result = (uint32_t)(lo - lo0);
    }
    p.toff[gid] = result;
}

// One workgroup per (document tile, query).  SQ counters of this kernel show the SIMDs' issue ports busy all the
// time (eight waves per SIMD, each issuing 13 % of its life), so the loop is written for few instructions per
// posting: a token's posting range is wave-uniform (scalar base pointers, 32-bit places: a tile holds at most
// BM_TILE postings of a token), rows stay 32-bit (below 2^32, checked at create), and a row outside the tile only
// sets a flag that is looked at once at the end.  The tokens are applied strictly one after the other, a barrier in
// between (the summation order of the reference), but nothing else waits in that order: posting range and idf of
// up to BM_TOK tokens are looked up together (one thread per token: term -> indptr / offsets / idf is a chain of
// dependent loads), and the first BM_PF postings per thread of token j + 1 are in flight while token j is added.
template <bool SCORES_ONLY>
__global__ __launch_bounds__(BM_NT) void bm25_tile_kernel(Bm25Params p) {
    __shared__ __attribute__((aligned(16))) double acc[BM_TILE];
    __shared__ __attribute__((aligned(16))) uint32_t sel_lds[RAG_SEL_LDS_WORDS];
    __shared__ unsigned long long kmin_s;
    __shared__ int count_s[2];
    __shared__ int base_s;
    __shared__ uint32_t tk_lo[BM_TOK], tk_n[BM_TOK];
    __shared__ double tk_w[BM_TOK];

    const int tid = threadIdx.x;
    const int tile_in_launch = blockIdx.x / p.n_queries;
    const int q = blockIdx.x - tile_in_launch * p.n_queries;
    const int tile = p.tile_base + tile_in_launch;
    const int64_t r0 = (int64_t)tile * BM_TILE;
    const uint32_t r0u = (uint32_t)r0;
    const int n_in = (int)min((int64_t)BM_TILE, p.n_docs - r0);
    unsigned long long tau = 0ull;                     // every thread reads the same word: one request per wave
    if (!SCORES_ONLY && p.share_tau) tau = __hip_atomic_load(&p.tau[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    {
        double2* a2 = reinterpret_cast<double2*>(acc);
#pragma unroll
        for (int i = 0; i < BM_TILE / 2 / BM_NT; ++i) a2[i * BM_NT + tid] = make_double2(0.0, 0.0);
    }
    if (tid == 0) {
        count_s[0] = 0;
        count_s[1] = 0;
        kmin_s = ~0ull;
    }

    int t_begin = p.qoff[q], t_end = p.qoff[q + 1];
    if (t_begin < 0 || t_end > p.n_terms_total) {     // offsets beyond the term array
        if (tid == 0) rag_flag(p.fault, RAG_FAULT_SEGMENT);
        t_begin = max(t_begin, 0);
        t_end = min(t_end, p.n_terms_total);
    }
    const int per = p.n_tiles + 1;
    bool bad = false;
    for (int c0 = t_begin; c0 < t_end; c0 += BM_TOK) {
        const int nt = min(BM_TOK, t_end - c0);
        if (c0 != t_begin) __syncthreads();           // the previous chunk's ranges have been read by everyone
        if (tid < nt) {
            const int ti = c0 + tid;
            const int term = p.qterms[ti];
            uint32_t lo = 0, n = 0;
            double w = 0.0;
            if (term >= 0 && term < p.vocab) {        // out of vocabulary: contributes nothing
                const uint32_t a = p.toff[(int64_t)ti * per + tile], b = p.toff[(int64_t)ti * per + tile + 1];
                lo = (uint32_t)(p.indptr[term] + a);   // postings are fewer than 2^32 (checked at create)
                n = b - a;
                w = p.idf[term];
            }
            tk_lo[tid] = lo;
            tk_n[tid] = n;
            tk_w[tid] = w;
        }
        __syncthreads();                               // (also orders the zeroing above before the first sums)
        uint32_t nr[BM_PF];
        double ni[BM_PF];
        auto fetch = [&](int j) {
            const uint32_t lo = __builtin_amdgcn_readfirstlane(tk_lo[j]);
            const uint32_t n = __builtin_amdgcn_readfirstlane(tk_n[j]);
            const uint32_t* rows_j = p.rows + lo;
            const double* imp_j = p.impact + lo;
#pragma unroll
            for (int u = 0; u < BM_PF; ++u) {
                const uint32_t idx = (uint32_t)tid + u * BM_NT;
                nr[u] = 0xFFFFFFFFu;                   // no posting in this place
                ni[u] = 0.0;
                if (idx < n) {
                    nr[u] = rows_j[idx];
                    ni[u] = imp_j[idx];
                }
            }
        };
        auto apply = [&](uint32_t row, double impact, double w) {
            const uint32_t r = row - r0u;
            if (r < (uint32_t)BM_TILE) {
                // score += idf * impact, two roundings, no contraction
                acc[r] = __dadd_rn(acc[r], __dmul_rn(w, impact));
            } else {
                bad |= row != 0xFFFFFFFFu;             // postings not sorted by row: index is corrupt
            }
        };
        fetch(0);
        for (int j = 0; j < nt; ++j) {
            const uint32_t lo = __builtin_amdgcn_readfirstlane(tk_lo[j]);
            const uint32_t n = __builtin_amdgcn_readfirstlane(tk_n[j]);
            const double w = tk_w[j];
            uint32_t cr[BM_PF];
            double ci[BM_PF];
#pragma unroll
            for (int u = 0; u < BM_PF; ++u) {
                cr[u] = nr[u];
                ci[u] = ni[u];
            }
            if (j + 1 < nt) fetch(j + 1);
#pragma unroll
            for (int u = 0; u < BM_PF; ++u) apply(cr[u], ci[u], w);
            if (n > (uint32_t)(BM_PF * BM_NT)) {       // a token most documents of the tile contain
                const uint32_t* rows_j = p.rows + lo;
                const double* imp_j = p.impact + lo;
                for (uint32_t idx = (uint32_t)tid + BM_PF * BM_NT; idx < n; idx += BM_NT) apply(rows_j[idx], imp_j[idx], w);
            }
            __syncthreads();
        }
    }
    if (t_begin >= t_end) __syncthreads();             // no token: the zeroing still has to be seen by everyone
    if (__ballot(bad) && (tid & 63) == 0) rag_flag(p.fault, RAG_FAULT_BM25_POSTING);

    if (SCORES_ONLY) {
        for (int i = tid; i < n_in; i += BM_NT) p.dense_out[r0 + i] = acc[i];
        return;
    }

    // ---- which documents of this tile reach the shared threshold?  Thread t holds documents t, t + 256, ...:
    // bit i of `mask` = document i * BM_NT + t passes.
    // (the key order is the order of the doubles themselves, so the comparison is done on them: one instruction
    // per document instead of the key conversion; key 0, "no threshold yet", lies below every score)
    constexpr int PER = BM_TILE / BM_NT;
    const double tau_d = tau ? key_to_f64(tau) : -__builtin_huge_val();
    uint32_t mask = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int d = i * BM_NT + tid;
        if (d < n_in && acc[d] >= tau_d) mask |= 1u << i;
    }
    {
        int mine = __popc(mask);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
        if ((tid & 63) == 0 && mine) atomicAdd(&count_s[0], mine);
    }
    __syncthreads();
    const int n_pass = count_s[0];

    // Reserve this tile's place in the query's candidate list: every document that passes, or the k best.
    const int n_emit = min(n_pass, p.k);
    if (n_emit == 0) return;
    if (tid == 0) base_s = atomicAdd(&p.cand_ctr[(size_t)q * BM_CTR_STRIDE], n_emit);
    unsigned long long kmin = ~0ull;
    if (n_pass > p.k) {
        auto get = [&](int i, uint64_t& key, uint32_t& row) {
            key = f64_to_key(acc[i]);
            row = (uint32_t)(r0 + i);
        };
        SelThr<uint64_t> thr = radix_select<uint64_t, BM_NT>(get, n_in, p.k, sel_lds, tid);
        mask = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int d = i * BM_NT + tid;
            if (d < n_in) {
                const uint64_t key = f64_to_key(acc[d]);
                if (sel_pass(thr, key, (uint32_t)(r0 + d))) {
                    mask |= 1u << i;
                    kmin = min(kmin, (unsigned long long)key);
                }
            }
        }
        if (kmin != ~0ull) atomicMin(&kmin_s, kmin);
    }
    __syncthreads();                                   // base_s
    const size_t slot = (size_t)q * p.n_tiles * p.k + base_s;
    if (mask) {
        int at = atomicAdd(&count_s[1], __popc(mask));
        while (mask) {
            const int i = __ffs(mask) - 1;
            mask &= mask - 1;
            const int d = i * BM_NT + tid;
            if (at < n_emit) {
                p.cand_key[slot + at] = f64_to_key(acc[d]);
                p.cand_row[slot + at] = (uint32_t)(r0 + d);
            }
            ++at;
        }
    }
    __syncthreads();
    const int n_written = count_s[1];
    if (n_written != n_emit) {
        // internal error: the reserved places must still hold defined entries (the lowest key, rows of
        // this tile that were not emitted cannot be named here, so the merge may flag the list too)
        if (tid == 0) rag_flag(p.fault, RAG_FAULT_BM25_LIST);
        for (int i = max(min(n_written, n_emit), 0) + tid; i < n_emit; i += BM_NT) {
            p.cand_key[slot + i] = 0ull;
            p.cand_row[slot + i] = (uint32_t)r0;
        }
    }
    // only an exact selection proves that k documents reach kmin
    if (tid == 0 && p.share_tau && n_pass > p.k && n_written == p.k) atomicMax(&p.tau[q], kmin_s);
}

// list counters, one cache line apart -> the dense array the merge reads
__global__ void bm25_gather_counts_kernel(const int32_t* __restrict__ ctr, int n_queries, int32_t* __restrict__ count) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n_queries) count[q] = ctr[(size_t)q * BM_CTR_STRIDE];
}

// The K-th best score of a merged sample of tiles is a lower bound of the final K-th score: raise the
// query's threshold to it (the tiles raise it with atomicMax too, hence the same here).
__global__ void bm25_tau_from_sample_kernel(const double* __restrict__ score, const int32_t* __restrict__ count,
                                            int n_queries, int k, unsigned long long* __restrict__ tau) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_queries) return;
    if (count[q] >= k) atomicMax(&tau[q], (unsigned long long)f64_to_key(score[(size_t)q * k + k - 1]));
}

}  // namespace

struct rag_bm25_s {
    int64_t* indptr = nullptr;
    uint32_t* rows = nullptr;
    double* impact = nullptr;
    double* idf = nullptr;
    bool owns = false;
    int64_t n_docs = 0, vocab = 0, nnz = 0, row_base = 0;
    int device = 0;                     // the device the postings live on (current device at create)
    DevBuf toff, tau, cand_key, cand_row, cand_count;
};

extern "C" int rag_bm25_create(const int64_t* indptr, const uint32_t* rows, const double* impact, const double* idf,
                               int64_t n_docs, int64_t vocab, int32_t mem_kind, int64_t row_base, rag_bm25_t* out) {
    RAG_REQUIRE(out != nullptr, RAG_EINVAL, "rag_bm25_create: out is null");
    *out = nullptr;
    RAG_REQUIRE(indptr && idf, RAG_EINVAL, "rag_bm25_create: null pointer");
    RAG_REQUIRE(n_docs >= 0 && vocab >= 0, RAG_EINVAL, "rag_bm25_create: negative size");
    RAG_REQUIRE(row_base >= 0 && row_base + n_docs < (1ll << 32), RAG_ELIMIT,
                "rag_bm25_create: global rows must stay below 2^32");
    RAG_REQUIRE(mem_kind >= RAG_MEM_HOST_COPY && mem_kind <= RAG_MEM_DEVICE_BORROW, RAG_EINVAL,
                "rag_bm25_create: unknown mem_kind %d", mem_kind);
    int64_t nnz = 0;
    if (mem_kind == RAG_MEM_HOST_COPY) {
        nnz = indptr[vocab];
    } else {
        RAG_HIP_TRY(hipMemcpy(&nnz, indptr + vocab, sizeof(int64_t), hipMemcpyDeviceToHost));
    }
    RAG_REQUIRE(nnz >= 0 && nnz < (1ll << 32), RAG_ELIMIT, "rag_bm25_create: %lld postings out of range", (long long)nnz);
    RAG_REQUIRE(nnz == 0 || (rows && impact), RAG_EINVAL, "rag_bm25_create: null postings");
    int device = 0;
    RAG_HIP_TRY(hipGetDevice(&device));
    rag_bm25_s* h = new rag_bm25_s();
    h->device = device;
    h->n_docs = n_docs;
    h->vocab = vocab;
    h->nnz = nnz;
    h->row_base = row_base;
    if (mem_kind == RAG_MEM_DEVICE_BORROW) {
        h->indptr = const_cast<int64_t*>(indptr);
        h->rows = const_cast<uint32_t*>(rows);
        h->impact = const_cast<double*>(impact);
        h->idf = const_cast<double*>(idf);
    } else {
        h->owns = true;
        const hipMemcpyKind kind = mem_kind == RAG_MEM_HOST_COPY ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
        auto up = [&](void** dst, const void* src, size_t bytes) -> int {
            size_t alloc = bytes ? bytes : 16;
            hipError_t e = hipMalloc(dst, alloc);
            if (e != hipSuccess) { rag_set_error("rag_bm25_create: hipMalloc(%zu) failed: %s", alloc, hipGetErrorString(e)); return RAG_ENOMEM; }
            if (bytes) {
                e = hipMemcpy(*dst, src, bytes, kind);
                if (e != hipSuccess) { rag_set_error("rag_bm25_create: hipMemcpy failed: %s", hipGetErrorString(e)); return RAG_EHIP; }
            }
            return RAG_OK;
        };
        int rc = up((void**)&h->indptr, indptr, (size_t)(vocab + 1) * sizeof(int64_t));
        if (!rc) rc = up((void**)&h->rows, rows, (size_t)nnz * sizeof(uint32_t));
        if (!rc) rc = up((void**)&h->impact, impact, (size_t)nnz * sizeof(double));
        if (!rc) rc = up((void**)&h->idf, idf, (size_t)vocab * sizeof(double));
        if (rc) {
            rag_bm25_destroy(h);
            return rc;
        }
    }
    *out = h;
    return RAG_OK;
}

extern "C" int rag_bm25_destroy(rag_bm25_t h) {
    if (!h) return RAG_OK;
    if (h->owns) {
        if (h->indptr) (void)hipFree(h->indptr);
        if (h->rows) (void)hipFree(h->rows);
        if (h->impact) (void)hipFree(h->impact);
        if (h->idf) (void)hipFree(h->idf);
    }
    h->toff.release();
    h->tau.release();
    h->cand_key.release();
    h->cand_row.release();
    h->cand_count.release();
    delete h;
    return RAG_OK;
}

extern "C" int64_t rag_bm25_docs(rag_bm25_t h) { return h ? h->n_docs : -1; }

namespace {

size_t bm25_toff_bytes(rag_bm25_t h, int n_terms_total) {
    return ((size_t)n_terms_total * ((size_t)ceil_div64(h->n_docs, BM_TILE) + 1) + 1) * sizeof(uint32_t);
}

// `toff` = device buffer of bm25_toff_bytes() bytes, or NULL: then the handle's own buffer is grown as needed
int bm25_prepare(rag_bm25_t h, Bm25Params& p, const int32_t* d_qterms, const int32_t* d_qoff, int n_queries,
                 int n_terms_total, hipStream_t stream, uint32_t* toff = nullptr) {
    p.indptr = h->indptr;
    p.rows = h->rows;
    p.impact = h->impact;
    p.idf = h->idf;
    p.n_docs = h->n_docs;
    p.vocab = h->vocab;
    p.fault = rag_fault_word();
    p.qterms = d_qterms;
    p.qoff = d_qoff;
    p.n_queries = n_queries;
    p.n_terms_total = n_terms_total;
    p.n_tiles = (int)ceil_div64(h->n_docs, BM_TILE);
    if (toff == nullptr) {
        int rc = h->toff.ensure(bm25_toff_bytes(h, n_terms_total));
        if (rc) return rc;
        toff = (uint32_t*)h->toff.p;
    }
    p.toff = toff;
    if (n_terms_total > 0) {
        const int64_t n = (int64_t)n_terms_total * (p.n_tiles + 1);
        bm25_offsets_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, stream>>>(p);
        RAG_HIP_TRY(hipGetLastError());
    }
    return RAG_OK;
}

}  // namespace

namespace {

// Device work buffers of one search: per-(term, tile) posting offsets, per-query thresholds, the
// per-query candidate lists and their counters.
struct Bm25Work {
    uint32_t* toff;
    unsigned long long* tau;
    uint64_t* cand_key;
    uint32_t* cand_row;
    int32_t* cand_count;
};
struct Bm25WorkBytes {
    size_t toff, tau, cand_key, cand_row, cand_count;
};
Bm25WorkBytes bm25_work_bytes(rag_bm25_t h, int n_queries, int n_terms_total, int k) {
    const size_t slots = (size_t)n_queries * (size_t)ceil_div64(h->n_docs, BM_TILE);
    return {bm25_toff_bytes(h, n_terms_total), (size_t)n_queries * sizeof(unsigned long long), slots * k * sizeof(uint64_t),
            slots * k * sizeof(uint32_t), (size_t)n_queries * (BM_CTR_STRIDE + 1) * sizeof(int32_t)};
}

int bm25_check_search_args(rag_bm25_t h, const int32_t* d_qterms, const int32_t* d_qoff, int32_t n_queries,
                           int32_t n_terms_total, int32_t k, double* d_out_score, int64_t* d_out_row,
                           int32_t* d_out_count) {
    RAG_REQUIRE(h != nullptr, RAG_EINVAL, "rag_bm25_search: null handle");
    RAG_REQUIRE(n_queries >= 0 && n_terms_total >= 0 && k >= 1 && k <= RAG_MAX_K, RAG_ELIMIT,
                "rag_bm25_search: need n_queries >= 0 and 1 <= k <= %d (got %d, %d)", RAG_MAX_K, n_queries, k);
    RAG_REQUIRE(n_queries == 0 || (d_qoff && d_out_score && d_out_row && d_out_count && (d_qterms || n_terms_total == 0)),
                RAG_EINVAL, "rag_bm25_search: null pointer");
    return RAG_OK;
}

int bm25_empty_result(int32_t n_queries, int32_t k, double* d_out_score, int64_t* d_out_row, int32_t* d_out_count,
                      hipStream_t stream) {
    RAG_HIP_TRY(hipMemsetAsync(d_out_count, 0, (size_t)n_queries * sizeof(int32_t), stream));
    RAG_HIP_TRY(hipMemsetAsync(d_out_row, 0xff, (size_t)n_queries * k * sizeof(int64_t), stream));
    RAG_HIP_TRY(hipMemsetAsync(d_out_score, 0, (size_t)n_queries * k * sizeof(double), stream));
    return RAG_OK;
}

// The search proper on given work buffers (sized by bm25_work_bytes); allocates nothing.
int bm25_search_on(rag_bm25_t h, const Bm25Work& w, const int32_t* d_qterms, const int32_t* d_qoff, int32_t n_queries,
                   int32_t n_terms_total, int32_t k, double* d_out_score, int64_t* d_out_row, int32_t* d_out_count,
                   hipStream_t stream) {
    Bm25Params p;
    memset(&p, 0, sizeof(p));
    int rc = bm25_prepare(h, p, d_qterms, d_qoff, n_queries, n_terms_total, stream, w.toff);
    if (rc) return rc;
    p.k = k;
    p.share_tau = getenv("RAG_BM25_NO_TAU") ? 0 : 1;
    p.tau = w.tau;
    p.cand_key = w.cand_key;
    p.cand_row = w.cand_row;
    p.cand_ctr = w.cand_count;                                       // [n_queries][BM_CTR_STRIDE], then the dense counts
    p.cand_count = w.cand_count + (size_t)n_queries * BM_CTR_STRIDE;
    RAG_HIP_TRY(hipMemsetAsync(p.tau, 0, (size_t)n_queries * sizeof(unsigned long long), stream));
    RAG_HIP_TRY(hipMemsetAsync(p.cand_ctr, 0, (size_t)n_queries * BM_CTR_STRIDE * sizeof(int32_t), stream));
    const int64_t n_blocks = (int64_t)p.n_tiles * n_queries;
    RAG_REQUIRE(n_blocks < (1ll << 31), RAG_ELIMIT, "rag_bm25_search: %lld workgroups exceed the grid limit",
                (long long)n_blocks);
    // The threshold the tiles share while they run is the best single-tile K-th score, far below the final
    // K-th score.  With many tiles, a sample of them is scored and merged first: its K-th score is a much
    // tighter bound, and the remaining tiles then emit a handful of candidates each instead of dozens.
    int sample = 0;
    // One tile in sixteen at most: the sample's K-th score then leaves about 16 K candidates in all.
    if (p.share_tau && p.n_tiles >= 64 && !getenv("RAG_BM25_NO_SAMPLE")) {
        // enough sample tiles that the candidates the sampled bound leaves, about k * n_tiles / sample, stay within
        // what the merge stages in LDS (10M documents, k = 100: 159 tiles, 33 ms per 1024 queries against 45 ms with 32)
        int cap = std::max(32, (int)ceil_div64((int64_t)k * p.n_tiles, 3072));
        if (const char* e = getenv("RAG_BM25_SAMPLE_MAX")) cap = atoi(e) > 0 ? atoi(e) : cap;      // tuning aid
        sample = std::min(cap, p.n_tiles / 16);
    }
    if (sample) {
        p.tile_base = 0;
        bm25_tile_kernel<false><<<(unsigned)((int64_t)sample * n_queries), BM_NT, 0, stream>>>(p);
        RAG_HIP_TRY(hipGetLastError());
        bm25_gather_counts_kernel<<<(n_queries + 255) / 256, 256, 0, stream>>>(p.cand_ctr, n_queries, p.cand_count);
        RAG_HIP_TRY(hipGetLastError());
        rc = launch_merge_cand64(p.cand_key, p.cand_row, p.cand_count, n_queries, (int64_t)p.n_tiles * k, k, h->row_base,
                                 d_out_score, d_out_row, d_out_count, p.fault, stream);
        if (rc) return rc;
        bm25_tau_from_sample_kernel<<<(n_queries + 255) / 256, 256, 0, stream>>>(d_out_score, d_out_count, n_queries, k, p.tau);
        RAG_HIP_TRY(hipGetLastError());
    }
    p.tile_base = sample;
    bm25_tile_kernel<false><<<(unsigned)((int64_t)(p.n_tiles - sample) * n_queries), BM_NT, 0, stream>>>(p);
    RAG_HIP_TRY(hipGetLastError());
    bm25_gather_counts_kernel<<<(n_queries + 255) / 256, 256, 0, stream>>>(p.cand_ctr, n_queries, p.cand_count);
    RAG_HIP_TRY(hipGetLastError());
    return launch_merge_cand64(p.cand_key, p.cand_row, p.cand_count, n_queries, (int64_t)p.n_tiles * k, k, h->row_base,
                               d_out_score, d_out_row, d_out_count, p.fault, stream);
}

}  // namespace

extern "C" int rag_bm25_search_dev(rag_bm25_t h, const int32_t* d_qterms, const int32_t* d_qoff, int32_t n_queries,
                                   int32_t n_terms_total, int32_t k, double* d_out_score, int64_t* d_out_row,
                                   int32_t* d_out_count, void* stream_) {
    int rc = bm25_check_search_args(h, d_qterms, d_qoff, n_queries, n_terms_total, k, d_out_score, d_out_row, d_out_count);
    if (rc) return rc;
    if (n_queries == 0) return RAG_OK;
    hipStream_t stream = (hipStream_t)stream_;
    if (h->n_docs == 0) return bm25_empty_result(n_queries, k, d_out_score, d_out_row, d_out_count, stream);
    const Bm25WorkBytes need = bm25_work_bytes(h, n_queries, n_terms_total, k);
    if ((rc = h->toff.ensure(need.toff))) return rc;
    if ((rc = h->tau.ensure(need.tau))) return rc;
    if ((rc = h->cand_key.ensure(need.cand_key))) return rc;
    if ((rc = h->cand_row.ensure(need.cand_row))) return rc;
    if ((rc = h->cand_count.ensure(need.cand_count))) return rc;
    const Bm25Work w = {(uint32_t*)h->toff.p, (unsigned long long*)h->tau.p, (uint64_t*)h->cand_key.p,
                        (uint32_t*)h->cand_row.p, (int32_t*)h->cand_count.p};
    return bm25_search_on(h, w, d_qterms, d_qoff, n_queries, n_terms_total, k, d_out_score, d_out_row, d_out_count, stream);
}

extern "C" int rag_bm25_reserve(rag_bm25_t h, int32_t max_queries, int32_t max_terms_total, int32_t max_k) {
    RAG_REQUIRE(h != nullptr, RAG_EINVAL, "rag_bm25_reserve: null handle");
    RAG_REQUIRE(max_queries >= 1 && max_terms_total >= 0 && max_k >= 1 && max_k <= RAG_MAX_K, RAG_ELIMIT,
                "rag_bm25_reserve: need max_queries >= 1, max_terms_total >= 0 and 1 <= max_k <= %d", RAG_MAX_K);
    if (h->n_docs == 0) return RAG_OK;
    const Bm25WorkBytes need = bm25_work_bytes(h, max_queries, max_terms_total, max_k);
    int rc;
    if ((rc = h->toff.ensure(need.toff))) return rc;
    if ((rc = h->tau.ensure(need.tau))) return rc;
    if ((rc = h->cand_key.ensure(need.cand_key))) return rc;
    if ((rc = h->cand_row.ensure(need.cand_row))) return rc;
    return h->cand_count.ensure(need.cand_count);
}

// ---- the same search for callers that cannot allocate -----------------------------------------------
// The work buffers live in a file-scope device array instead of the handle's grow-only buffers: nothing in
// this entry point calls hipMalloc / hipFree or any routine that does.  Each GPU the library runs on holds
// its own copy of the array, at its own address, until the process ends; the address is looked up once per
// device (rag_bm25_static_setup, called by rag_init; without rag_init on the device the entry point refuses).
// Every call on a GPU uses that GPU's one array: calls from different streams or threads are chained on the
// device with an event (created by rag_init), so they run one after the other; only replays of captured graphs
// that overlap other searches on the same GPU can still race.  The array keeps the last call's contents; the thresholds and list counters, which must
// start at zero, are cleared by every call.
constexpr size_t BM_STATIC_BYTES = 32u << 20;
static __device__ __align__(256) unsigned char scratch[BM_STATIC_BYTES];

namespace {
constexpr int BM_MAX_DEVICES = 64;
std::mutex g_static_mu;
unsigned char* g_static_ptr[BM_MAX_DEVICES] = {};   // device address of `scratch` on each device
hipEvent_t g_static_done[BM_MAX_DEVICES] = {};      // recorded behind the last search that used the device's array
bool g_static_used[BM_MAX_DEVICES] = {};
}  // namespace

// Look up this device's address of the static work array (the array's name itself is only a host-side
// stand-in and must never be handed to a kernel or a memset).
int rag_bm25_static_setup() {
    int dev = 0;
    RAG_HIP_TRY(hipGetDevice(&dev));
    RAG_REQUIRE(dev >= 0 && dev < BM_MAX_DEVICES, RAG_ELIMIT, "device index %d beyond %d", dev, BM_MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_static_mu);
    if (g_static_ptr[dev] == nullptr) {
        unsigned char* ptr = nullptr;
        RAG_HIP_TRY(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
        RAG_REQUIRE(ptr != nullptr, RAG_EHIP, "hipGetSymbolAddress returned a null address for the static work array");
        hipEvent_t ev = nullptr;
        RAG_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        g_static_done[dev] = ev;
        g_static_ptr[dev] = ptr;
    }
    return RAG_OK;
}

extern "C" int64_t rag_bm25_search_static_capacity(void) { return (int64_t)BM_STATIC_BYTES; }

extern "C" int rag_bm25_search_static_dev(rag_bm25_t h, const int32_t* d_qterms, const int32_t* d_qoff, int32_t n_queries,
                                          int32_t n_terms_total, int32_t k, double* d_out_score, int64_t* d_out_row,
                                          int32_t* d_out_count, void* stream_) {
    int rc = bm25_check_search_args(h, d_qterms, d_qoff, n_queries, n_terms_total, k, d_out_score, d_out_row, d_out_count);
    if (rc) return rc;
    if (n_queries == 0) return RAG_OK;
    hipStream_t stream = (hipStream_t)stream_;
    if (h->n_docs == 0) return bm25_empty_result(n_queries, k, d_out_score, d_out_row, d_out_count, stream);
    // carve the array: every piece starts on a 256-byte boundary
    const Bm25WorkBytes need = bm25_work_bytes(h, n_queries, n_terms_total, k);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_toff = 0, o_tau = o_toff + up(need.toff), o_key = o_tau + up(need.tau),
                 o_row = o_key + up(need.cand_key), o_cnt = o_row + up(need.cand_row), total = o_cnt + up(need.cand_count);
    RAG_REQUIRE(total <= BM_STATIC_BYTES, RAG_ELIMIT,
                "rag_bm25_search_static: %d queries with %d terms, k = %d over %lld documents need %zu bytes of work "
                "space, the static array holds %zu (use rag_bm25_search_dev, or fewer queries per call)",
                n_queries, n_terms_total, k, (long long)h->n_docs, total, (size_t)BM_STATIC_BYTES);
    int dev = 0;
    RAG_HIP_TRY(hipGetDevice(&dev));
    RAG_REQUIRE(dev >= 0 && dev < BM_MAX_DEVICES, RAG_ELIMIT, "device index %d beyond %d", dev, BM_MAX_DEVICES);
    RAG_REQUIRE(dev == h->device, RAG_EINVAL,
                "rag_bm25_search_static_dev: the index lives on device %d, the current device is %d (each GPU has its own "
                "work array)", h->device, dev);
    unsigned char* ptr;
    {
        std::lock_guard<std::mutex> lock(g_static_mu);
        ptr = g_static_ptr[dev];
    }
    // rag_init(device) looks up the array's address and allocates the device's fault word; doing either here would be
    // the allocation (and the synchronous call) this entry point promises never to make
    RAG_REQUIRE(ptr != nullptr && rag_fault_word_ready(), RAG_EINVAL,
                "rag_bm25_search_static_dev: call rag_init(%d) first (it sets up this device's static work array and "
                "fault word; this entry point allocates nothing)", dev);
    const Bm25Work w = {(uint32_t*)(ptr + o_toff), (unsigned long long*)(ptr + o_tau), (uint64_t*)(ptr + o_key),
                        (uint32_t*)(ptr + o_row), (int32_t*)(ptr + o_cnt)};
    // Calls on different streams (or threads) share the device's one array: each search first waits, on the device,
    // for the search before it, and leaves an event behind its own kernels.  The lock covers wait + launches + record,
    // so the chain is the order in which the calls took it.  A capturing stream is left alone (an event recorded
    // outside the capture cannot be waited for inside it): a captured search is ordered by its graph, and replays that
    // overlap other searches on the same GPU remain the caller's to serialise.
    std::lock_guard<std::mutex> lock(g_static_mu);
    const bool chained = !rag_stream_capturing(stream);
    if (chained && g_static_used[dev]) RAG_HIP_TRY(hipStreamWaitEvent(stream, g_static_done[dev], 0));
    rc = bm25_search_on(h, w, d_qterms, d_qoff, n_queries, n_terms_total, k, d_out_score, d_out_row, d_out_count, stream);
    if (chained) {
        RAG_HIP_TRY(hipEventRecord(g_static_done[dev], stream));
        g_static_used[dev] = true;
    }
    return rc;
}

#define BM_TRY_OR_CLEAN(expr)                                                                    \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            rag_set_error("%s failed: %s", #expr, hipGetErrorString(_e));                        \
            cleanup();                                                                           \
            return RAG_EHIP;                                                                     \
        }                                                                                        \
    } while (0)

extern "C" int rag_bm25_search(rag_bm25_t h, const int32_t* qterms, const int32_t* qoff, int32_t n_queries, int32_t k,
                               double* out_score, int64_t* out_row, int32_t* out_count) {
    RAG_REQUIRE(h != nullptr, RAG_EINVAL, "rag_bm25_search: null handle");
    RAG_REQUIRE(n_queries >= 0 && k >= 1 && k <= RAG_MAX_K, RAG_ELIMIT,
                "rag_bm25_search: need n_queries >= 0 and 1 <= k <= %d (got %d, %d)", RAG_MAX_K, n_queries, k);
    if (n_queries == 0) return RAG_OK;
    RAG_REQUIRE(qoff && out_score && out_row && out_count, RAG_EINVAL, "rag_bm25_search: null pointer");
    const int n_terms = qoff[n_queries];
    RAG_REQUIRE(qoff[0] == 0 && n_terms >= 0 && (qterms || n_terms == 0), RAG_EINVAL, "rag_bm25_search: bad query offsets");
    const size_t nk = (size_t)n_queries * k;
    int32_t *d_t = nullptr, *d_o = nullptr, *d_c = nullptr;
    double* d_s = nullptr;
    int64_t* d_r = nullptr;
    auto cleanup = [&]() {
        if (d_t) (void)hipFree(d_t);
        if (d_o) (void)hipFree(d_o);
        if (d_c) (void)hipFree(d_c);
        if (d_s) (void)hipFree(d_s);
        if (d_r) (void)hipFree(d_r);
    };
    BM_TRY_OR_CLEAN(hipMalloc((void**)&d_t, (size_t)(n_terms + 1) * sizeof(int32_t)));
    BM_TRY_OR_CLEAN(hipMalloc((void**)&d_o, (size_t)(n_queries + 1) * sizeof(int32_t)));
    BM_TRY_OR_CLEAN(hipMalloc((void**)&d_c, (size_t)n_queries * sizeof(int32_t)));
    BM_TRY_OR_CLEAN(hipMalloc((void**)&d_s, nk * sizeof(double)));
    BM_TRY_OR_CLEAN(hipMalloc((void**)&d_r, nk * sizeof(int64_t)));
    if (n_terms) BM_TRY_OR_CLEAN(hipMemcpy(d_t, qterms, (size_t)n_terms * sizeof(int32_t), hipMemcpyHostToDevice));
    BM_TRY_OR_CLEAN(hipMemcpy(d_o, qoff, (size_t)(n_queries + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    int rc = rag_bm25_search_dev(h, d_t, d_o, n_queries, n_terms, k, d_s, d_r, d_c, nullptr);
    if (rc) {
        cleanup();
        return rc;
    }
    BM_TRY_OR_CLEAN(hipMemcpy(out_score, d_s, nk * sizeof(double), hipMemcpyDeviceToHost));
    BM_TRY_OR_CLEAN(hipMemcpy(out_row, d_r, nk * sizeof(int64_t), hipMemcpyDeviceToHost));
    BM_TRY_OR_CLEAN(hipMemcpy(out_count, d_c, (size_t)n_queries * sizeof(int32_t), hipMemcpyDeviceToHost));
    cleanup();
    return RAG_OK;
}

extern "C" int rag_bm25_scores(rag_bm25_t h, const int32_t* qterms, int32_t n_terms, double* out_scores) {
    RAG_REQUIRE(h != nullptr, RAG_EINVAL, "rag_bm25_scores: null handle");
    RAG_REQUIRE(n_terms >= 0 && (qterms || n_terms == 0) && out_scores, RAG_EINVAL, "rag_bm25_scores: bad argument");
    if (h->n_docs == 0) return RAG_OK;
    int32_t *d_t = nullptr, *d_o = nullptr;
    double* d_s = nullptr;
    auto cleanup = [&]() {
        if (d_t) (void)hipFree(d_t);
        if (d_o) (void)hipFree(d_o);
        if (d_s) (void)hipFree(d_s);
    };
    const int32_t qoff[2] = {0, n_terms};
    BM_TRY_OR_CLEAN(hipMalloc((void**)&d_t, (size_t)(n_terms + 1) * sizeof(int32_t)));
    BM_TRY_OR_CLEAN(hipMalloc((void**)&d_o, 2 * sizeof(int32_t)));
    BM_TRY_OR_CLEAN(hipMalloc((void**)&d_s, (size_t)h->n_docs * sizeof(double)));
    if (n_terms) BM_TRY_OR_CLEAN(hipMemcpy(d_t, qterms, (size_t)n_terms * sizeof(int32_t), hipMemcpyHostToDevice));
    BM_TRY_OR_CLEAN(hipMemcpy(d_o, qoff, sizeof(qoff), hipMemcpyHostToDevice));
    Bm25Params p;
    memset(&p, 0, sizeof(p));
    int rc = bm25_prepare(h, p, d_t, d_o, 1, n_terms, nullptr);
    if (rc) {
        cleanup();
        return rc;
    }
    p.dense_out = d_s;
    bm25_tile_kernel<true><<<(unsigned)p.n_tiles, BM_NT>>>(p);
    BM_TRY_OR_CLEAN(hipGetLastError());
    BM_TRY_OR_CLEAN(hipMemcpy(out_scores, d_s, (size_t)h->n_docs * sizeof(double), hipMemcpyDeviceToHost));
    cleanup();
    return RAG_OK;
}
