// Per-query merge of several best-first candidate lists into one sorted top-K list.
// One 256-thread workgroup per query: radix-select the K best (key, row) pairs, gather
// them into LDS, order them by rank counting, write (score, global row).
#pragma once

#include "common.h"

// internal linkage: every translation unit that includes this header gets kernels of its own
namespace ragmerge {
namespace {

constexpr int MERGE_NT = 256;

// ---- item sources: get(q, i, key, row) -> item is valid --------------------------------
struct Cand32Src {                 // dense scan output: (key, local row) pairs + counts
    const uint2* cand;
    const int32_t* count;
    int n_lists, k_in;
    typedef uint32_t key_t;
    static constexpr bool dense = false;
    __device__ int size(int) const { return n_lists * k_in; }
    __device__ bool get(int q, int i, uint32_t& key, uint32_t& row) const {
        int list = i / k_in, j = i - list * k_in;
        if (j >= count[(size_t)q * n_lists + list]) { key = 0; row = 0; return false; }
        uint2 v = cand[((size_t)q * n_lists + list) * k_in + j];
        key = v.x; row = v.y;
        return true;
    }
};
struct Cand64Src {                 // BM25 scan output: one append-only list per query
    const uint64_t* key;           // [n_queries][capacity]
    const uint32_t* row;
    const int32_t* count;          // [n_queries] entries appended so far
    int64_t capacity;
    typedef uint64_t key_t;
    // Every entry below size(q) is valid, and the scan's threshold keeps the list short: the merge copies it
    // into LDS and selects there.
    static constexpr bool dense = true;
    __device__ int size(int q) const { return count[q]; }
    __device__ bool get(int q, int i, uint64_t& k, uint32_t& r) const {
        size_t at = (size_t)q * capacity + i;
        k = key[at]; r = row[at];
        return true;
    }
};
struct ListF32Src {                // public merge: fp32 scores + int64 global rows
    const float* score;
    const int64_t* rowp;
    const int32_t* count;          // may be null: then row < 0 marks an unused slot
    int n_lists, k_in, invert;
    typedef uint32_t key_t;
    static constexpr bool dense = false;
    __device__ int size(int) const { return n_lists * k_in; }
    __device__ bool get(int q, int i, uint32_t& key, uint32_t& row) const {
        int list = i / k_in, j = i - list * k_in;
        size_t at = ((size_t)q * n_lists + list) * k_in + j;
        bool ok = count ? j < count[(size_t)q * n_lists + list] : true;
        int64_t r = ok ? rowp[at] : -1;
        if (r < 0) { key = 0; row = 0; return false; }
        uint32_t kk = f32_to_key(score[at]);
        key = invert ? ~kk : kk; row = (uint32_t)r;
        return true;
    }
};
struct ListF64Src {
    const double* score;
    const int64_t* rowp;
    const int32_t* count;
    int n_lists, k_in;
    typedef uint64_t key_t;
    static constexpr bool dense = false;
    __device__ int size(int) const { return n_lists * k_in; }
    __device__ bool get(int q, int i, uint64_t& key, uint32_t& row) const {
        int list = i / k_in, j = i - list * k_in;
        size_t at = ((size_t)q * n_lists + list) * k_in + j;
        bool ok = count ? j < count[(size_t)q * n_lists + list] : true;
        int64_t r = ok ? rowp[at] : -1;
        if (r < 0) { key = 0; row = 0; return false; }
        key = f64_to_key(score[at]); row = (uint32_t)r;
        return true;
    }
};

template <typename KeyT> struct ScoreOf;
template <> struct ScoreOf<uint32_t> {
    typedef float type;
    __device__ static float from(uint32_t k, int invert) { return key_to_f32(invert ? ~k : k); }
    __device__ static float empty() { return 0.f; }
};
template <> struct ScoreOf<uint64_t> {
    typedef double type;
    __device__ static double from(uint64_t k, int) { return key_to_f64(k); }
    __device__ static double empty() { return 0.0; }
};

// Entries a compactable source may hold for the merge to stage them in LDS (12 bytes each for 64-bit keys).
constexpr int MERGE_COMPACT = 4096;

// Select the K best of n items, order them, write (score, global row).  Every thread of the block calls.
template <typename KeyT, typename Get>
__device__ __forceinline__ void select_and_emit(Get get, int n, int K, int64_t row_base, int invert,
                                                typename ScoreOf<KeyT>::type* os, int64_t* orow, int32_t* out_count_q,
                                                uint32_t* fault, uint32_t* sel_lds, KeyT* sel_key, uint32_t* sel_row,
                                                int* gathered, int tid) {
    auto load = [&](int i, KeyT& key, uint32_t& row) { get(i, key, row); };
    SelThr<KeyT> thr = radix_select<KeyT, MERGE_NT>(load, n, K, sel_lds, tid);

    for (int i = tid; i < n; i += MERGE_NT) {
        KeyT key; uint32_t row;
        bool ok = get(i, key, row);
        if (ok && sel_pass(thr, key, row)) {
            int at = atomicAdd(gathered, 1);
            if (at < RAG_MAX_K) { sel_key[at] = key; sel_row[at] = row; }
        }
    }
    __syncthreads();
    // the selection yields exactly K entries; anything else is an internal error: report it
    // and emit only entries that were really gathered
    const int got = min(*gathered, RAG_MAX_K);
    const int kept = min(got, K);
    if (got != K) {
        if (tid == 0) {
            rag_flag(fault, RAG_FAULT_MERGE_SHORT);
            *out_count_q = kept;
        }
        for (int i = kept + tid; i < K; i += MERGE_NT) { os[i] = ScoreOf<KeyT>::empty(); orow[i] = -1; }
    }
    // rank counting: position = number of selected items that are better
    for (int i = tid; i < kept; i += MERGE_NT) {
        const KeyT key = sel_key[i];
        const uint32_t row = sel_row[i];
        int rank = 0;
        for (int j = 0; j < kept; ++j) {
            KeyT kj = sel_key[j];
            uint32_t rj = sel_row[j];
            rank += (kj > key || (kj == key && rj > row)) ? 1 : 0;
        }
        os[rank] = ScoreOf<KeyT>::from(key, invert);
        orow[rank] = (int64_t)row + row_base;
    }
}

template <typename Src>
__global__ __launch_bounds__(MERGE_NT) void merge_topk_kernel(Src src, int k_out, int64_t row_base, int invert,
                                                              typename ScoreOf<typename Src::key_t>::type* out_score,
                                                              int64_t* out_row, int32_t* out_count,
                                                              uint32_t* fault) {
    typedef typename Src::key_t KeyT;
    typedef typename ScoreOf<KeyT>::type ScoreT;
    constexpr int STAGE = Src::dense ? MERGE_COMPACT : 1;
    __shared__ __attribute__((aligned(16))) uint32_t sel_lds[RAG_SEL_LDS_WORDS];
    __shared__ KeyT sel_key[RAG_MAX_K];
    __shared__ uint32_t sel_row[RAG_MAX_K];
    __shared__ KeyT stage_key[STAGE];
    __shared__ uint32_t stage_row[STAGE];
    __shared__ int counter[2];
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int n_items = src.size(q);

    if (tid < 2) counter[tid] = 0;
    __syncthreads();
    int n_valid = n_items;
    if constexpr (!Src::dense) {
        int local_valid = 0;
        for (int i = tid; i < n_items; i += MERGE_NT) {
            KeyT key; uint32_t row;
            local_valid += src.get(q, i, key, row) ? 1 : 0;
        }
        if (local_valid) atomicAdd(&counter[0], local_valid);
        __syncthreads();
        n_valid = counter[0];
    }
    const int K = min(k_out, n_valid);

    ScoreT* os = out_score + (size_t)q * k_out;
    int64_t* orow = out_row + (size_t)q * k_out;
    for (int i = K + tid; i < k_out; i += MERGE_NT) { os[i] = ScoreOf<KeyT>::empty(); orow[i] = -1; }
    if (tid == 0) out_count[q] = K;
    if (K == 0) return;

    if constexpr (Src::dense) {
        if (n_items <= MERGE_COMPACT) {
            for (int i = tid; i < n_items; i += MERGE_NT) {
                KeyT key; uint32_t row;
                src.get(q, i, key, row);
                stage_key[i] = key;
                stage_row[i] = row;
            }
            __syncthreads();
            auto get = [&](int i, KeyT& key, uint32_t& row) { key = stage_key[i]; row = stage_row[i]; return true; };
            select_and_emit<KeyT>(get, n_items, K, row_base, invert, os, orow, out_count + q, fault, sel_lds, sel_key,
                                  sel_row, &counter[1], tid);
            return;
        }
    }
    auto get = [&](int i, KeyT& key, uint32_t& row) { return src.get(q, i, key, row); };
    select_and_emit<KeyT>(get, n_items, K, row_base, invert, os, orow, out_count + q, fault, sel_lds, sel_key, sel_row,
                          &counter[1], tid);
}

}  // namespace
}  // namespace ragmerge

static inline int launch_merge_cand32(const uint2* cand, const int32_t* count, int n_queries, int n_lists, int k_in,
                                      int k_out, int64_t row_base, float* out_score, int64_t* out_row,
                                      int32_t* out_count, uint32_t* fault, hipStream_t stream) {
    ragmerge::Cand32Src src{cand, count, n_lists, k_in};
    hipLaunchKernelGGL(ragmerge::merge_topk_kernel<ragmerge::Cand32Src>, dim3(n_queries), dim3(ragmerge::MERGE_NT), 0,
                       stream, src, k_out, row_base, 0, out_score, out_row, out_count, fault);
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

static inline int launch_merge_cand64(const uint64_t* key, const uint32_t* row, const int32_t* count, int n_queries,
                                      int64_t capacity, int k_out, int64_t row_base, double* out_score,
                                      int64_t* out_row, int32_t* out_count, uint32_t* fault, hipStream_t stream) {
    ragmerge::Cand64Src src{key, row, count, capacity};
    hipLaunchKernelGGL(ragmerge::merge_topk_kernel<ragmerge::Cand64Src>, dim3(n_queries), dim3(ragmerge::MERGE_NT), 0,
                       stream, src, k_out, row_base, 0, out_score, out_row, out_count, fault);
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}
