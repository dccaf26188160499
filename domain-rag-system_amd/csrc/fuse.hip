// Rank fusion (RRF / weighted), segmented rerank top-k, cross-encoder pair packing and the
// public list merge.  One wavefront per query: the lists are a few hundred entries, so the
// work is wave-wide compares and reductions over LDS-resident lists.  gfx950 only.
//
// Replaces HybridIndexer._reciprocal_rank_fusion / _weighted_fusion (reference
// src/indexing/hybrid_indexer.py:112-218), the sort + cut of BaseReranker.rerank (reference
// src/reranker/rerank.py:42-43) and the pair batching of CrossEncoder.predict behind
// CrossEncoderReranker.score_pairs (reference src/reranker/rerank.py:79-84).
#include "common.h"
#include "merge.h"

#include <mutex>
#include <vector>

namespace {

constexpr int FUSE_WAVES = 2;                      // queries per workgroup (LDS: 21 KB each)
constexpr int FUSE_MAX = 2 * RAG_MAX_FUSE_LIST;    // entries of both lists together

struct FuseParams {
    const int64_t* vec_row;  const double* vec_score;  const int32_t* vec_count;  int vec_stride;
    const int64_t* bm_row;   const double* bm_score;  const int32_t* bm_count;   int bm_stride;
    const uint32_t* canon;
    int64_t n_canon;
    uint32_t* fault;
    int n_queries, top_k;
    double k_const, vw, bw;
    int32_t* out_pos;  int64_t* out_row;  double* out_score;  int32_t* out_count;
};

// ids[e]: fusion key of entry e (entries: the vector list, then the BM25 list).
// val[e]: the entry's contribution.  Python's dict semantics, restated per unique id:
//   total(id) = ((0 + val[e1]) + val[e2]) + ...   over its entries in list order,
//   ids ordered by first appearance; output = stable sort of the ids by total, descending.
// TEAM = waves that share one query.  Large batches run one wave per query (FUSE_WAVES queries per workgroup: a
// thousand queries fill the chip); for a handful of queries the three quadratic passes of a single wave are the
// latency of the whole stage (153 us for one query with two lists of 100), so a workgroup of four waves takes one
// query, a thread an entry: every thread still sums its own id's entries in list order, so the totals are the same bits.
template <bool WEIGHTED, int TEAM>
__global__ __launch_bounds__(FUSE_WAVES * 64 > TEAM * 64 ? FUSE_WAVES * 64 : TEAM * 64) void fuse_kernel(FuseParams p) {
    constexpr int SLOTS = TEAM == 1 ? FUSE_WAVES : 1;      // queries per workgroup
    constexpr int NT = 64 * TEAM;                          // threads per query
    __shared__ uint32_t ids_s[SLOTS][FUSE_MAX];
    __shared__ double val_s[SLOTS][FUSE_MAX];
    __shared__ double tot_s[SLOTS][FUSE_MAX];   // total of the id, at its first entry; NaN-free marker below
    __shared__ uint8_t lead_s[SLOTS][FUSE_MAX];
    __shared__ double red_s[4][TEAM];
    __shared__ int cnt_s[TEAM];
    const int lane = threadIdx.x % NT, w = threadIdx.x / NT, wave = (threadIdx.x >> 6) % TEAM;
    const int q = blockIdx.x * SLOTS + w;
    if (q >= p.n_queries) return;
    uint32_t* ids = ids_s[w];
    double* val = val_s[w];
    double* tot = tot_s[w];
    uint8_t* lead = lead_s[w];
    auto team_sync = [&]() {
        if constexpr (TEAM == 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            __syncthreads();
        }
    };

    const int nv = min(p.vec_count ? p.vec_count[q] : 0, p.vec_stride);
    const int nb = min(p.bm_count ? p.bm_count[q] : 0, p.bm_stride);
    const int n = nv + nb;
    const int64_t* vrow = p.vec_row + (size_t)q * p.vec_stride;
    const int64_t* brow = p.bm_row + (size_t)q * p.bm_stride;

    double vmin = 0, vmax = 0, bmin = 0, bmax = 0;
    if (WEIGHTED) {
        // per-list min / max (exact, order free)
        double lo = INFINITY, hi = -INFINITY, lo2 = INFINITY, hi2 = -INFINITY;
        for (int e = lane; e < nv; e += NT) { double s = (double)p.vec_score[(size_t)q * p.vec_stride + e]; lo = fmin(lo, s); hi = fmax(hi, s); }
        for (int e = lane; e < nb; e += NT) { double s = p.bm_score[(size_t)q * p.bm_stride + e]; lo2 = fmin(lo2, s); hi2 = fmax(hi2, s); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo = fmin(lo, __shfl_xor(lo, off, 64)); hi = fmax(hi, __shfl_xor(hi, off, 64));
            lo2 = fmin(lo2, __shfl_xor(lo2, off, 64)); hi2 = fmax(hi2, __shfl_xor(hi2, off, 64));
        }
        if constexpr (TEAM > 1) {
            if ((threadIdx.x & 63) == 0) { red_s[0][wave] = lo; red_s[1][wave] = hi; red_s[2][wave] = lo2; red_s[3][wave] = hi2; }
            __syncthreads();
            for (int t = 0; t < TEAM; ++t) {
                lo = fmin(lo, red_s[0][t]); hi = fmax(hi, red_s[1][t]); lo2 = fmin(lo2, red_s[2][t]); hi2 = fmax(hi2, red_s[3][t]);
            }
        }
        vmin = lo; vmax = hi; bmin = lo2; bmax = hi2;
    }

    for (int e = lane; e < n; e += NT) {
        const bool is_vec = e < nv;
        const int64_t row = is_vec ? vrow[e] : brow[e - nv];
        uint32_t id = (uint32_t)row;
        if (p.canon != nullptr) {
            if (row >= 0 && row < p.n_canon) id = p.canon[row];
            else rag_flag(p.fault, RAG_FAULT_FUSE_ROW);     // fuses on its own id
        }
        ids[e] = id;
        double c;
        if (!WEIGHTED) {
            const int rank = (is_vec ? e : e - nv) + 1;
            c = __ddiv_rn(1.0, __dadd_rn(p.k_const, (double)rank));
        } else if (is_vec) {
            double s = (double)p.vec_score[(size_t)q * p.vec_stride + e];
            if (vmax != vmin) s = __ddiv_rn(__dsub_rn(s, vmin), __dsub_rn(vmax, vmin));
            c = __dmul_rn(p.vw, s);
        } else {
            double s = p.bm_score[(size_t)q * p.bm_stride + (e - nv)];
            if (bmax != bmin) s = __ddiv_rn(__dsub_rn(s, bmin), __dsub_rn(bmax, bmin));
            c = __dmul_rn(p.bw, s);
        }
        val[e] = c;
    }
    team_sync();

    // first appearance + in-order totals
    int mine = 0;
    for (int e = lane; e < n; e += NT) {
        const uint32_t id = ids[e];
        bool first = true;
        for (int j = 0; j < e; ++j) if (ids[j] == id) { first = false; break; }
        lead[e] = first ? 1 : 0;
        mine += first ? 1 : 0;
        if (first) {
            double t = val[e];                      // 0 + x == x
            for (int j = e + 1; j < n; ++j) if (ids[j] == id) t = __dadd_rn(t, val[j]);
            tot[e] = t;
        }
    }
    // rank among the unique ids: better total first, ties by first appearance
    int n_unique = mine;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n_unique += __shfl_xor(n_unique, off, 64);
    if constexpr (TEAM > 1) {
        if ((threadIdx.x & 63) == 0) cnt_s[wave] = n_unique;
    }
    team_sync();
    if constexpr (TEAM > 1) {
        n_unique = 0;
        for (int t = 0; t < TEAM; ++t) n_unique += cnt_s[t];
    }
    const int n_out = min(n_unique, p.top_k);
    int32_t* opos = p.out_pos + (size_t)q * p.top_k;
    int64_t* orow = p.out_row + (size_t)q * p.top_k;
    double* osc = p.out_score + (size_t)q * p.top_k;
    for (int i = n_out + lane; i < p.top_k; i += NT) { opos[i] = -1; orow[i] = -1; osc[i] = 0.0; }
    if (lane == 0) p.out_count[q] = n_out;

    for (int e = lane; e < n; e += NT) {
        if (!lead[e]) continue;
        const double t = tot[e];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            if (!lead[j]) continue;
            const double tj = tot[j];
            rank += (tj > t || (tj == t && j < e)) ? 1 : 0;
        }
        if (rank < n_out) {
            // which entry supplies the payload
            int pos = e;                            // RRF: the first appearance
            if (WEIGHTED) {
                // a vector entry always (re)claims the id: the last vector entry wins;
                // otherwise the first BM25 entry
                const uint32_t id = ids[e];
                for (int j = nv - 1; j >= 0; --j) if (ids[j] == id) { pos = j; break; }
            }
            opos[rank] = pos;
            orow[rank] = pos < nv ? vrow[pos] : brow[pos - nv];
            osc[rank] = t;
        }
    }
}

constexpr int FUSE_TEAM = 4;                       // waves per query of the small-batch form
constexpr int FUSE_TEAM_MAX_QUERIES = 128;         // ... used up to this many queries

template <bool WEIGHTED>
void launch_fuse(const FuseParams& p, int n_queries, hipStream_t stream) {
    if (n_queries <= FUSE_TEAM_MAX_QUERIES)
        fuse_kernel<WEIGHTED, FUSE_TEAM><<<n_queries, FUSE_TEAM * 64, 0, stream>>>(p);
    else
        fuse_kernel<WEIGHTED, 1><<<(n_queries + FUSE_WAVES - 1) / FUSE_WAVES, FUSE_WAVES * 64, 0, stream>>>(p);
}

// Stable descending order of each pool's scores, cut at top_k: one wave per pool.
__global__ __launch_bounds__(FUSE_WAVES * 64) void topk_segmented_kernel(const float* __restrict__ score,
                                                                         int n_scores,
                                                                         const int32_t* __restrict__ seg_off,
                                                                         int n_queries, int top_k,
                                                                         int32_t* __restrict__ out_idx,
                                                                         int32_t* __restrict__ out_count,
                                                                         uint32_t* fault) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * FUSE_WAVES + w;
    if (q >= n_queries) return;
    const int b = seg_off[q];
    int n = seg_off[q + 1] - b;
    if (b < 0 || n < 0 || (int64_t)b + n > n_scores) {   // pool outside the score array: empty
        if (lane == 0) rag_flag(fault, RAG_FAULT_SEGMENT);
        n = 0;
    }
    const int n_out = min(n, top_k);
    int32_t* out = out_idx + (size_t)q * top_k;
    for (int i = n_out + lane; i < top_k; i += 64) out[i] = -1;
    if (lane == 0) out_count[q] = n_out;
    for (int i = lane; i < n; i += 64) {
        const float s = score[b + i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const float sj = score[b + j];
            rank += (sj > s || (sj == s && j < i)) ? 1 : 0;
        }
        if (rank < n_out) out[rank] = i;
    }
}

// One wave per pair: [CLS] query [SEP] chunk [SEP], padded to seq_len.
__global__ __launch_bounds__(256) void ce_pack_kernel(const int32_t* __restrict__ query_tok,
                                                      const int64_t* __restrict__ query_off, int n_queries,
                                                      const int32_t* __restrict__ chunk_tok,
                                                      const int64_t* __restrict__ chunk_off, int64_t n_chunks,
                                                      int64_t chunk_row_base,
                                                      const int32_t* __restrict__ pair_query,
                                                      const int64_t* __restrict__ pair_row,
                                                      const int32_t* __restrict__ pair_slot, int n_pairs, int n_slots,
                                                      int seq_len, int cls_id, int sep_id, int pad_id,
                                                      int64_t* __restrict__ ids, int64_t* __restrict__ types,
                                                      int64_t* __restrict__ mask, uint32_t* fault) {
    const int lane = threadIdx.x & 63;
    const int pr = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pr >= n_pairs) return;
    const int64_t slot = pair_slot ? pair_slot[pr] : pr;
    if (slot < 0 || slot >= n_slots) {                   // nowhere to write this pair
        if (lane == 0) rag_flag(fault, RAG_FAULT_CE_PAIR);
        return;
    }
    int64_t* oi = ids + slot * seq_len;
    int64_t* ot = types + slot * seq_len;
    int64_t* om = mask + slot * seq_len;
    const int64_t row = pair_row[pr];
    const int64_t lrow = row - chunk_row_base;
    const int qi = pair_query[pr];
    const bool bad = row >= 0 && (lrow < 0 || lrow >= n_chunks || qi < 0 || qi >= n_queries);
    if (bad && lane == 0) rag_flag(fault, RAG_FAULT_CE_PAIR);
    if (row < 0 || bad) {
        for (int i = lane; i < seq_len; i += 64) { oi[i] = pad_id; ot[i] = 0; om[i] = 0; }
        return;
    }
    const int64_t qb = query_off[qi];
    int qn = (int)(query_off[qi + 1] - qb);
    const int64_t cb = chunk_off[lrow];
    int cn = (int)(chunk_off[lrow + 1] - cb);
    qn = max(qn, 0);
    cn = max(cn, 0);
    qn = min(qn, max(seq_len - 3, 0));
    cn = min(cn, max(seq_len - 3 - qn, 0));
    const int first = 2 + qn;               // [CLS] q.. [SEP]
    const int total = first + cn + 1;       // .. c.. [SEP]
    for (int i = lane; i < seq_len; i += 64) {
        int64_t tok;
        int64_t tt = 0, m = 1;
        if (i == 0) tok = cls_id;
        else if (i < 1 + qn) tok = query_tok[qb + i - 1];
        else if (i == 1 + qn) tok = sep_id;
        else if (i < first + cn) { tok = chunk_tok[cb + i - first]; tt = 1; }
        else if (i == first + cn) { tok = sep_id; tt = 1; }
        else { tok = pad_id; m = 0; }
        if (i >= total) { tok = pad_id; tt = 0; m = 0; }
        oi[i] = tok; ot[i] = tt; om[i] = m;
    }
}

// One wave per pair, packed (unpadded) output.
__global__ __launch_bounds__(256) void ce_pack_packed_kernel(const int32_t* __restrict__ query_tok,
                                                             const int64_t* __restrict__ query_off, int n_queries,
                                                             const int32_t* __restrict__ chunk_tok,
                                                             const int64_t* __restrict__ chunk_off, int64_t n_chunks,
                                                             int64_t chunk_row_base,
                                                             const int32_t* __restrict__ pair_query,
                                                             const int64_t* __restrict__ pair_row,
                                                             const int32_t* __restrict__ cu, int n_pairs,
                                                             int64_t total_tokens, int max_len, int cls_id, int sep_id,
                                                             int32_t* __restrict__ ids, int32_t* __restrict__ types,
                                                             int32_t* __restrict__ pos, uint32_t* fault) {
    const int lane = threadIdx.x & 63;
    const int pr = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pr >= n_pairs) return;
    const int64_t start = cu[pr];
    const int64_t n = (int64_t)cu[pr + 1] - start;
    if (n == 0) return;
    const int64_t row = pair_row[pr];
    const int64_t lrow = row - chunk_row_base;
    const int qi = pair_query[pr];
    bool bad = start < 0 || n < 0 || start + n > total_tokens || row < 0 || lrow < 0 || lrow >= n_chunks || qi < 0 ||
               qi >= n_queries;
    int qn = 0, cn = 0;
    int64_t qb = 0, cb = 0;
    if (!bad) {
        qb = query_off[qi];
        cb = chunk_off[lrow];
        qn = max((int)(query_off[qi + 1] - qb), 0);
        cn = max((int)(chunk_off[lrow + 1] - cb), 0);
        qn = min(qn, max(max_len - 3, 0));
        cn = min(cn, max(max_len - 3 - qn, 0));
        bad = qn + cn + 3 != n;                          // cu was not built from these lengths
    }
    if (bad) {
        if (lane == 0) rag_flag(fault, RAG_FAULT_CE_PAIR);
        return;
    }
    const int first = 2 + qn;                            // [CLS] q.. [SEP]
    for (int i = lane; i < (int)n; i += 64) {
        int tok, tt = 0;
        if (i == 0) tok = cls_id;
        else if (i < 1 + qn) tok = query_tok[qb + i - 1];
        else if (i == 1 + qn) tok = sep_id;
        else if (i < first + cn) { tok = chunk_tok[cb + i - first]; tt = 1; }
        else { tok = sep_id; tt = 1; }
        ids[start + i] = tok;
        types[start + i] = tt;
        pos[start + i] = i;
    }
}

__global__ void ce_pair_len_kernel(const int64_t* __restrict__ query_off, int n_queries,
                                   const int64_t* __restrict__ chunk_off, int64_t n_chunks,
                                   int64_t chunk_row_base, const int32_t* __restrict__ pair_query,
                                   const int64_t* __restrict__ pair_row, int n_pairs, int max_len,
                                   int32_t* __restrict__ out_len, uint32_t* fault) {
    const int pr = blockIdx.x * blockDim.x + threadIdx.x;
    if (pr >= n_pairs) return;
    const int64_t row = pair_row[pr];
    if (row < 0) { out_len[pr] = 0; return; }
    const int64_t lrow = row - chunk_row_base;
    const int qi = pair_query[pr];
    if (lrow < 0 || lrow >= n_chunks || qi < 0 || qi >= n_queries) {
        rag_flag(fault, RAG_FAULT_CE_PAIR);
        out_len[pr] = 0;
        return;
    }
    int qn = max((int)(query_off[qi + 1] - query_off[qi]), 0);
    int cn = max((int)(chunk_off[lrow + 1] - chunk_off[lrow]), 0);
    qn = min(qn, max(max_len - 3, 0));
    cn = min(cn, max(max_len - 3 - qn, 0));
    out_len[pr] = qn + cn + 3;
}

int check_fuse_args(const char* who, const void* vec_row, const void* bm_row, int vec_stride, int bm_stride,
                    int n_queries, int top_k, const void* a, const void* b, const void* c, const void* d) {
    RAG_REQUIRE(n_queries >= 0 && top_k >= 1 && top_k <= RAG_MAX_K, RAG_ELIMIT, "%s: need 1 <= top_k <= %d", who, RAG_MAX_K);
    RAG_REQUIRE(vec_stride >= 0 && vec_stride <= RAG_MAX_FUSE_LIST && bm_stride >= 0 && bm_stride <= RAG_MAX_FUSE_LIST,
                RAG_ELIMIT, "%s: list strides must be within [0, %d]", who, RAG_MAX_FUSE_LIST);
    RAG_REQUIRE((vec_row || vec_stride == 0) && (bm_row || bm_stride == 0), RAG_EINVAL, "%s: null list", who);
    RAG_REQUIRE(a && b && c && d, RAG_EINVAL, "%s: null output", who);
    return RAG_OK;
}

}  // namespace

extern "C" int rag_fuse_rrf_dev(const int64_t* d_vec_row, const int32_t* d_vec_count, int32_t vec_stride,
                                const int64_t* d_bm_row, const int32_t* d_bm_count, int32_t bm_stride,
                                const uint32_t* d_canon, int64_t n_canon, int32_t n_queries, int32_t top_k,
                                double k_const, int32_t* d_out_pos, int64_t* d_out_row, double* d_out_score,
                                int32_t* d_out_count, void* stream) {
    int rc = check_fuse_args("rag_fuse_rrf", d_vec_row, d_bm_row, vec_stride, bm_stride, n_queries, top_k, d_out_pos,
                             d_out_row, d_out_score, d_out_count);
    if (rc) return rc;
    if (n_queries == 0) return RAG_OK;
    FuseParams p;
    memset(&p, 0, sizeof(p));
    p.vec_row = d_vec_row; p.vec_count = vec_stride ? d_vec_count : nullptr; p.vec_stride = vec_stride;
    p.bm_row = d_bm_row; p.bm_count = bm_stride ? d_bm_count : nullptr; p.bm_stride = bm_stride;
    p.canon = n_canon > 0 ? d_canon : nullptr; p.n_canon = n_canon; p.fault = rag_fault_word();
    p.n_queries = n_queries; p.top_k = top_k; p.k_const = k_const;
    p.out_pos = d_out_pos; p.out_row = d_out_row; p.out_score = d_out_score; p.out_count = d_out_count;
    launch_fuse<false>(p, n_queries, (hipStream_t)stream);
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

extern "C" int rag_fuse_weighted_dev(const int64_t* d_vec_row, const double* d_vec_score, const int32_t* d_vec_count,
                                     int32_t vec_stride, const int64_t* d_bm_row, const double* d_bm_score,
                                     const int32_t* d_bm_count, int32_t bm_stride, const uint32_t* d_canon,
                                     int64_t n_canon, int32_t n_queries, int32_t top_k, double vector_weight,
                                     double bm25_weight,
                                     int32_t* d_out_pos, int64_t* d_out_row, double* d_out_score,
                                     int32_t* d_out_count, void* stream) {
    int rc = check_fuse_args("rag_fuse_weighted", d_vec_row, d_bm_row, vec_stride, bm_stride, n_queries, top_k,
                             d_out_pos, d_out_row, d_out_score, d_out_count);
    if (rc) return rc;
    RAG_REQUIRE((d_vec_score || vec_stride == 0) && (d_bm_score || bm_stride == 0), RAG_EINVAL,
                "rag_fuse_weighted: null scores");
    if (n_queries == 0) return RAG_OK;
    FuseParams p;
    memset(&p, 0, sizeof(p));
    p.vec_row = d_vec_row; p.vec_score = d_vec_score; p.vec_count = vec_stride ? d_vec_count : nullptr; p.vec_stride = vec_stride;
    p.bm_row = d_bm_row; p.bm_score = d_bm_score; p.bm_count = bm_stride ? d_bm_count : nullptr; p.bm_stride = bm_stride;
    p.canon = n_canon > 0 ? d_canon : nullptr; p.n_canon = n_canon; p.fault = rag_fault_word();
    p.n_queries = n_queries; p.top_k = top_k; p.vw = vector_weight; p.bw = bm25_weight;
    p.out_pos = d_out_pos; p.out_row = d_out_row; p.out_score = d_out_score; p.out_count = d_out_count;
    launch_fuse<true>(p, n_queries, (hipStream_t)stream);
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

// ---- host-pointer forms -------------------------------------------------------------------
namespace {

struct Staging {
    void* ptrs[16];
    int n = 0;
    ~Staging() { for (int i = 0; i < n; ++i) if (ptrs[i]) (void)hipFree(ptrs[i]); }
    // device copy of `bytes` of host memory (or an uninitialised buffer when src is null)
    int up(void** dst, const void* src, size_t bytes) {
        *dst = nullptr;
        size_t alloc = bytes ? bytes : 16;
        hipError_t e = hipMalloc(dst, alloc);
        if (e != hipSuccess) { rag_set_error("hipMalloc(%zu) failed: %s", alloc, hipGetErrorString(e)); return RAG_ENOMEM; }
        ptrs[n++] = *dst;
        if (src && bytes) {
            e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
            if (e != hipSuccess) { rag_set_error("hipMemcpy failed: %s", hipGetErrorString(e)); return RAG_EHIP; }
        }
        return RAG_OK;
    }
    int down(void* dst, const void* src, size_t bytes) {
        if (!bytes) return RAG_OK;
        hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { rag_set_error("hipMemcpy failed: %s", hipGetErrorString(e)); return RAG_EHIP; }
        return RAG_OK;
    }
};

int fuse_host(bool weighted, const int64_t* vec_row, const double* vec_score, const int32_t* vec_count,
              int32_t vec_stride, const int64_t* bm_row, const double* bm_score, const int32_t* bm_count,
              int32_t bm_stride, const uint32_t* canon, int64_t n_canon, int32_t n_queries, int32_t top_k, double a,
              double b, int32_t* out_pos, int64_t* out_row, double* out_score, int32_t* out_count) {
    if (n_queries <= 0) return n_queries == 0 ? RAG_OK : RAG_EINVAL;
    Staging st;
    int64_t *d_vr, *d_br, *d_or;
    double *d_vs, *d_bs, *d_os;
    int32_t *d_vc, *d_bc, *d_op, *d_oc;
    uint32_t* d_canon = nullptr;
    int rc;
    const size_t nq = (size_t)n_queries;
    if ((rc = st.up((void**)&d_vr, vec_row, nq * vec_stride * sizeof(int64_t)))) return rc;
    if ((rc = st.up((void**)&d_br, bm_row, nq * bm_stride * sizeof(int64_t)))) return rc;
    if ((rc = st.up((void**)&d_vs, weighted ? vec_score : nullptr, nq * vec_stride * sizeof(double)))) return rc;
    if ((rc = st.up((void**)&d_bs, weighted ? bm_score : nullptr, nq * bm_stride * sizeof(double)))) return rc;
    if ((rc = st.up((void**)&d_vc, vec_stride ? vec_count : nullptr, nq * sizeof(int32_t)))) return rc;
    if ((rc = st.up((void**)&d_bc, bm_stride ? bm_count : nullptr, nq * sizeof(int32_t)))) return rc;
    if (canon && n_canon > 0 && (rc = st.up((void**)&d_canon, canon, (size_t)n_canon * sizeof(uint32_t)))) return rc;
    if ((rc = st.up((void**)&d_op, nullptr, nq * top_k * sizeof(int32_t)))) return rc;
    if ((rc = st.up((void**)&d_or, nullptr, nq * top_k * sizeof(int64_t)))) return rc;
    if ((rc = st.up((void**)&d_os, nullptr, nq * top_k * sizeof(double)))) return rc;
    if ((rc = st.up((void**)&d_oc, nullptr, nq * sizeof(int32_t)))) return rc;
    if (weighted)
        rc = rag_fuse_weighted_dev(d_vr, d_vs, d_vc, vec_stride, d_br, d_bs, d_bc, bm_stride, d_canon,
                                   d_canon ? n_canon : 0, n_queries, top_k, a, b, d_op, d_or, d_os, d_oc, nullptr);
    else
        rc = rag_fuse_rrf_dev(d_vr, d_vc, vec_stride, d_br, d_bc, bm_stride, d_canon, d_canon ? n_canon : 0, n_queries,
                              top_k, a, d_op, d_or, d_os, d_oc, nullptr);
    if (rc) return rc;
    if ((rc = st.down(out_pos, d_op, nq * top_k * sizeof(int32_t)))) return rc;
    if ((rc = st.down(out_row, d_or, nq * top_k * sizeof(int64_t)))) return rc;
    if ((rc = st.down(out_score, d_os, nq * top_k * sizeof(double)))) return rc;
    return st.down(out_count, d_oc, nq * sizeof(int32_t));
}

}  // namespace

extern "C" int rag_fuse_rrf(const int64_t* vec_row, const int32_t* vec_count, int32_t vec_stride,
                            const int64_t* bm_row, const int32_t* bm_count, int32_t bm_stride, const uint32_t* canon,
                            int64_t n_canon, int32_t n_queries, int32_t top_k, double k_const, int32_t* out_pos,
                            int64_t* out_row, double* out_score, int32_t* out_count) {
    int rc = check_fuse_args("rag_fuse_rrf", vec_row, bm_row, vec_stride, bm_stride, n_queries, top_k, out_pos, out_row,
                             out_score, out_count);
    if (rc) return rc;
    return fuse_host(false, vec_row, nullptr, vec_count, vec_stride, bm_row, nullptr, bm_count, bm_stride, canon,
                     n_canon, n_queries, top_k, k_const, 0.0, out_pos, out_row, out_score, out_count);
}

extern "C" int rag_fuse_weighted(const int64_t* vec_row, const double* vec_score, const int32_t* vec_count,
                                 int32_t vec_stride, const int64_t* bm_row, const double* bm_score,
                                 const int32_t* bm_count, int32_t bm_stride, const uint32_t* canon, int64_t n_canon,
                                 int32_t n_queries, int32_t top_k, double vector_weight, double bm25_weight,
                                 int32_t* out_pos, int64_t* out_row, double* out_score, int32_t* out_count) {
    int rc = check_fuse_args("rag_fuse_weighted", vec_row, bm_row, vec_stride, bm_stride, n_queries, top_k, out_pos,
                             out_row, out_score, out_count);
    if (rc) return rc;
    return fuse_host(true, vec_row, vec_score, vec_count, vec_stride, bm_row, bm_score, bm_count, bm_stride, canon,
                     n_canon, n_queries, top_k, vector_weight, bm25_weight, out_pos, out_row, out_score, out_count);
}

extern "C" int rag_topk_segmented_dev(const float* d_score, int32_t n_scores, const int32_t* d_seg_off,
                                      int32_t n_queries, int32_t top_k, int32_t* d_out_idx, int32_t* d_out_count,
                                      void* stream) {
    RAG_REQUIRE(n_queries >= 0 && n_scores >= 0 && top_k >= 1 && top_k <= RAG_MAX_K, RAG_ELIMIT,
                "rag_topk_segmented: need 1 <= top_k <= %d", RAG_MAX_K);
    if (n_queries == 0) return RAG_OK;
    RAG_REQUIRE((d_score || n_scores == 0) && d_seg_off && d_out_idx && d_out_count, RAG_EINVAL,
                "rag_topk_segmented: null pointer");
    topk_segmented_kernel<<<(n_queries + FUSE_WAVES - 1) / FUSE_WAVES, FUSE_WAVES * 64, 0, (hipStream_t)stream>>>(
        d_score, n_scores, d_seg_off, n_queries, top_k, d_out_idx, d_out_count, rag_fault_word());
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

extern "C" int rag_topk_segmented(const float* score, const int32_t* seg_off, int32_t n_queries, int32_t top_k,
                                  int32_t* out_idx, int32_t* out_count) {
    RAG_REQUIRE(n_queries >= 0 && top_k >= 1 && top_k <= RAG_MAX_K, RAG_ELIMIT, "rag_topk_segmented: need 1 <= top_k <= %d",
                RAG_MAX_K);
    if (n_queries == 0) return RAG_OK;
    RAG_REQUIRE(score && seg_off && out_idx && out_count, RAG_EINVAL, "rag_topk_segmented: null pointer");
    const int n = seg_off[n_queries];
    RAG_REQUIRE(n >= 0, RAG_EINVAL, "rag_topk_segmented: bad offsets");
    Staging st;
    float* d_s;
    int32_t *d_o, *d_i, *d_c;
    int rc;
    if ((rc = st.up((void**)&d_s, score, (size_t)n * sizeof(float)))) return rc;
    if ((rc = st.up((void**)&d_o, seg_off, (size_t)(n_queries + 1) * sizeof(int32_t)))) return rc;
    if ((rc = st.up((void**)&d_i, nullptr, (size_t)n_queries * top_k * sizeof(int32_t)))) return rc;
    if ((rc = st.up((void**)&d_c, nullptr, (size_t)n_queries * sizeof(int32_t)))) return rc;
    if ((rc = rag_topk_segmented_dev(d_s, n, d_o, n_queries, top_k, d_i, d_c, nullptr))) return rc;
    if ((rc = st.down(out_idx, d_i, (size_t)n_queries * top_k * sizeof(int32_t)))) return rc;
    return st.down(out_count, d_c, (size_t)n_queries * sizeof(int32_t));
}

extern "C" int rag_ce_pack_dev(const int32_t* d_query_tok, const int64_t* d_query_off, int32_t n_queries,
                               const int32_t* d_chunk_tok, const int64_t* d_chunk_off, int64_t n_chunks,
                               int64_t chunk_row_base, const int32_t* d_pair_query, const int64_t* d_pair_row,
                               const int32_t* d_pair_slot, int32_t n_pairs, int32_t n_slots, int32_t seq_len,
                               int32_t cls_id, int32_t sep_id, int32_t pad_id, int64_t* d_input_ids,
                               int64_t* d_token_type_ids, int64_t* d_attention_mask, void* stream) {
    RAG_REQUIRE(n_pairs >= 0 && seq_len >= 3, RAG_EINVAL, "rag_ce_pack: need n_pairs >= 0 and seq_len >= 3");
    RAG_REQUIRE(n_queries >= 0 && n_chunks >= 0 && n_slots >= 0 && (d_pair_slot || n_slots >= n_pairs), RAG_EINVAL,
                "rag_ce_pack: negative size, or fewer slots than pairs");
    if (n_pairs == 0) return RAG_OK;
    RAG_REQUIRE(d_query_tok && d_query_off && d_chunk_tok && d_chunk_off && d_pair_query && d_pair_row && d_input_ids &&
                    d_token_type_ids && d_attention_mask,
                RAG_EINVAL, "rag_ce_pack: null pointer");
    ce_pack_kernel<<<(n_pairs + 3) / 4, 256, 0, (hipStream_t)stream>>>(
        d_query_tok, d_query_off, n_queries, d_chunk_tok, d_chunk_off, n_chunks, chunk_row_base, d_pair_query,
        d_pair_row, d_pair_slot, n_pairs, n_slots, seq_len, cls_id, sep_id, pad_id, d_input_ids, d_token_type_ids,
        d_attention_mask, rag_fault_word());
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

extern "C" int rag_ce_pack_packed_dev(const int32_t* d_query_tok, const int64_t* d_query_off, int32_t n_queries,
                                      const int32_t* d_chunk_tok, const int64_t* d_chunk_off, int64_t n_chunks,
                                      int64_t chunk_row_base, const int32_t* d_pair_query, const int64_t* d_pair_row,
                                      const int32_t* d_cu, int32_t n_pairs, int64_t total_tokens, int32_t max_len,
                                      int32_t cls_id, int32_t sep_id, int32_t* d_input_ids, int32_t* d_token_type_ids,
                                      int32_t* d_positions, void* stream) {
    RAG_REQUIRE(n_pairs >= 0 && max_len >= 3 && total_tokens >= 0 && n_queries >= 0 && n_chunks >= 0, RAG_EINVAL,
                "rag_ce_pack_packed: bad sizes");
    if (n_pairs == 0 || total_tokens == 0) return RAG_OK;
    RAG_REQUIRE(d_query_tok && d_query_off && d_chunk_tok && d_chunk_off && d_pair_query && d_pair_row && d_cu &&
                    d_input_ids && d_token_type_ids && d_positions,
                RAG_EINVAL, "rag_ce_pack_packed: null pointer");
    ce_pack_packed_kernel<<<(n_pairs + 3) / 4, 256, 0, (hipStream_t)stream>>>(
        d_query_tok, d_query_off, n_queries, d_chunk_tok, d_chunk_off, n_chunks, chunk_row_base, d_pair_query,
        d_pair_row, d_cu, n_pairs, total_tokens, max_len, cls_id, sep_id, d_input_ids, d_token_type_ids, d_positions,
        rag_fault_word());
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

extern "C" int rag_ce_pair_len_dev(const int64_t* d_query_off, int32_t n_queries, const int64_t* d_chunk_off,
                                   int64_t n_chunks, int64_t chunk_row_base, const int32_t* d_pair_query,
                                   const int64_t* d_pair_row, int32_t n_pairs, int32_t max_len, int32_t* d_out_len,
                                   void* stream) {
    RAG_REQUIRE(n_pairs >= 0 && max_len >= 3, RAG_EINVAL, "rag_ce_pair_len: need n_pairs >= 0 and max_len >= 3");
    RAG_REQUIRE(n_queries >= 0 && n_chunks >= 0, RAG_EINVAL, "rag_ce_pair_len: negative size");
    if (n_pairs == 0) return RAG_OK;
    RAG_REQUIRE(d_query_off && d_chunk_off && d_pair_query && d_pair_row && d_out_len, RAG_EINVAL,
                "rag_ce_pair_len: null pointer");
    ce_pair_len_kernel<<<(n_pairs + 255) / 256, 256, 0, (hipStream_t)stream>>>(
        d_query_off, n_queries, d_chunk_off, n_chunks, chunk_row_base, d_pair_query, d_pair_row, n_pairs, max_len,
        d_out_len, rag_fault_word());
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

// ---- public merges ---------------------------------------------------------------------------
extern "C" int rag_merge_topk_f32_dev(const float* d_score, const int64_t* d_row, const int32_t* d_count,
                                      int32_t n_queries, int32_t n_lists, int32_t k_in, int32_t k_out,
                                      int32_t smaller_is_better, float* d_out_score, int64_t* d_out_row,
                                      int32_t* d_out_count, void* stream) {
    RAG_REQUIRE(n_queries >= 0 && n_lists >= 1 && k_in >= 1 && k_out >= 1 && k_out <= RAG_MAX_K, RAG_ELIMIT,
                "rag_merge_topk: bad sizes");
    if (n_queries == 0) return RAG_OK;
    RAG_REQUIRE(d_score && d_row && d_out_score && d_out_row && d_out_count, RAG_EINVAL, "rag_merge_topk: null pointer");
    ragmerge::ListF32Src src{d_score, d_row, d_count, n_lists, k_in, smaller_is_better ? 1 : 0};
    hipLaunchKernelGGL(ragmerge::merge_topk_kernel<ragmerge::ListF32Src>, dim3(n_queries), dim3(ragmerge::MERGE_NT), 0,
                       (hipStream_t)stream, src, k_out, (int64_t)0, smaller_is_better ? 1 : 0,
                       d_out_score, d_out_row, d_out_count, rag_fault_word());
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

extern "C" int rag_merge_topk_f64_dev(const double* d_score, const int64_t* d_row, const int32_t* d_count,
                                      int32_t n_queries, int32_t n_lists, int32_t k_in, int32_t k_out,
                                      double* d_out_score, int64_t* d_out_row, int32_t* d_out_count, void* stream) {
    RAG_REQUIRE(n_queries >= 0 && n_lists >= 1 && k_in >= 1 && k_out >= 1 && k_out <= RAG_MAX_K, RAG_ELIMIT,
                "rag_merge_topk: bad sizes");
    if (n_queries == 0) return RAG_OK;
    RAG_REQUIRE(d_score && d_row && d_out_score && d_out_row && d_out_count, RAG_EINVAL, "rag_merge_topk: null pointer");
    ragmerge::ListF64Src src{d_score, d_row, d_count, n_lists, k_in};
    hipLaunchKernelGGL(ragmerge::merge_topk_kernel<ragmerge::ListF64Src>, dim3(n_queries), dim3(ragmerge::MERGE_NT), 0,
                       (hipStream_t)stream, src, k_out, (int64_t)0, 0, d_out_score, d_out_row,
                       d_out_count, rag_fault_word());
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

// ---- library ----------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void rag_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* rag_last_error(void) { return g_err; }
// ---- payload filters: per-row categorical codes -> row mask ----------------------------------------
namespace {

constexpr int FILTER_MAX_FIELDS = 8;
struct FilterParams {
    const int32_t* codes;              // [n_fields][n_rows]
    int64_t n_rows;
    int n_fields;
    int32_t want[FILTER_MAX_FIELDS];   // code a row must carry in that field; RAG_FILTER_ANY = unconstrained
    uint8_t* mask;
};

// mask[row] = 1 iff every constrained field of the row carries the wanted code.  Four rows per thread:
// the constrained columns are read as 16-byte pieces, unconstrained ones not at all.
__global__ void filter_mask_kernel(FilterParams p) {
    const int64_t r0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (r0 >= p.n_rows) return;
    bool keep[4] = {true, true, true, true};
    const bool whole = r0 + 4 <= p.n_rows && (p.n_rows & 3) == 0;
    for (int f = 0; f < p.n_fields; ++f) {
        if (p.want[f] == RAG_FILTER_ANY) continue;
        const int32_t* col = p.codes + (int64_t)f * p.n_rows;
        if (whole) {
            const int4 c = *reinterpret_cast<const int4*>(col + r0);
            keep[0] = keep[0] && c.x == p.want[f];
            keep[1] = keep[1] && c.y == p.want[f];
            keep[2] = keep[2] && c.z == p.want[f];
            keep[3] = keep[3] && c.w == p.want[f];
        } else {
            for (int j = 0; j < 4; ++j)
                if (r0 + j < p.n_rows) keep[j] = keep[j] && col[r0 + j] == p.want[f];
        }
    }
    for (int j = 0; j < 4; ++j)
        if (r0 + j < p.n_rows) p.mask[r0 + j] = keep[j] ? 1 : 0;
}

}  // namespace

extern "C" int rag_filter_mask_dev(const int32_t* d_codes, int32_t n_fields, int64_t n_rows, const int32_t* want,
                                   uint8_t* d_mask, void* stream) {
    RAG_REQUIRE(n_fields >= 0 && n_fields <= FILTER_MAX_FIELDS, RAG_ELIMIT, "rag_filter_mask: at most %d fields (got %d)",
                FILTER_MAX_FIELDS, n_fields);
    RAG_REQUIRE(n_rows >= 0, RAG_EINVAL, "rag_filter_mask: negative row count");
    if (n_rows == 0) return RAG_OK;
    RAG_REQUIRE(d_mask && (n_fields == 0 || (d_codes && want)), RAG_EINVAL, "rag_filter_mask: null pointer");
    FilterParams p;
    p.codes = d_codes; p.n_rows = n_rows; p.n_fields = n_fields; p.mask = d_mask;
    for (int f = 0; f < FILTER_MAX_FIELDS; ++f) p.want[f] = f < n_fields ? want[f] : RAG_FILTER_ANY;
    hipLaunchKernelGGL(filter_mask_kernel, dim3((unsigned)ceil_div64(ceil_div64(n_rows, 4), 256)), dim3(256), 0,
                       (hipStream_t)stream, p);
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

extern "C" int rag_version(void) { return 101; }

// one fault word per device, allocated on first use and kept for the life of the process
namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_fault_mu;
uint32_t* g_fault[MAX_DEVICES] = {};
}  // namespace

uint32_t* rag_fault_word() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(g_fault_mu);
    if (g_fault[dev] == nullptr) {
        uint32_t* w = nullptr;
        if (hipMalloc((void**)&w, 256) != hipSuccess) return nullptr;
        if (hipMemset(w, 0, 256) != hipSuccess) {
            (void)hipFree(w);
            return nullptr;
        }
        g_fault[dev] = w;
    }
    return g_fault[dev];
}

// whether the current device's fault word exists already (never allocates)
bool rag_fault_word_ready() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return false;
    std::lock_guard<std::mutex> lock(g_fault_mu);
    return g_fault[dev] != nullptr;
}

// dynamic-LDS grants, per (device, kernel)
namespace {
struct LdsGrant { const void* fn; size_t bytes; };
std::mutex g_grant_mu;
std::vector<LdsGrant> g_grants[MAX_DEVICES];
}  // namespace

int rag_grant_dyn_lds(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return RAG_OK;                 // within the default limit
    int dev = 0;
    RAG_HIP_TRY(hipGetDevice(&dev));
    RAG_REQUIRE(dev >= 0 && dev < MAX_DEVICES, RAG_ELIMIT, "device index %d beyond %d", dev, MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_grant_mu);
    LdsGrant* g = nullptr;
    for (auto& e : g_grants[dev])
        if (e.fn == kernel) g = &e;
    if (g != nullptr && g->bytes >= bytes) return RAG_OK;
    RAG_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    if (g != nullptr) g->bytes = bytes;
    else g_grants[dev].push_back({kernel, bytes});
    return RAG_OK;
}

extern "C" int rag_fault_flags(uint32_t* out_flags, int32_t clear) {
    RAG_REQUIRE(out_flags != nullptr, RAG_EINVAL, "rag_fault_flags: null pointer");
    uint32_t* w = rag_fault_word();
    RAG_REQUIRE(w != nullptr, RAG_ENOMEM, "rag_fault_flags: the fault word could not be allocated");
    RAG_HIP_TRY(hipDeviceSynchronize());
    RAG_HIP_TRY(hipMemcpy(out_flags, w, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (clear) RAG_HIP_TRY(hipMemset(w, 0, sizeof(uint32_t)));
    return RAG_OK;
}

extern "C" int rag_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        rag_set_error("rag_init: no HIP device (%s)", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
        return RAG_ENODEV;
    }
    RAG_REQUIRE(device >= 0 && device < n, RAG_EINVAL, "rag_init: device %d out of range (0..%d)", device, n - 1);
    RAG_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    RAG_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        rag_set_error("rag_init: device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
        return RAG_ENODEV;
    }
    RAG_REQUIRE(rag_fault_word() != nullptr, RAG_ENOMEM, "rag_init: the fault word could not be allocated");
    return rag_bm25_static_setup();          // this device's address of the no-allocation search's work array
}
