// BM25 index build on the device: tokenised documents (term ids, document-major) -> term-major
// CSR postings (indptr, rows, tf) and their fp64 impacts.  gfx950 only.
//
// Behind BM25Indexer.index_chunks (reference src/indexing/bm25_indexer.py:80-111), which hands the
// token lists to rank_bm25's BM25Okapi: that constructor counts, per document, the frequency of
// every distinct term (a dict per document) and the number of documents each term occurs in.  The
// same counts here, laid out for the search kernel:
//   1. one wave per document marks the first occurrence of every term in it and counts the term's
//      occurrences (the document's tokens pass through LDS 64 at a time; quadratic in the document
//      length, which is tens of tokens for <= 512-character chunks);
//   2. an exclusive scan of the per-document distinct-term counts gives every document its slice of
//      the posting stream, so the stream is document-major;
//   3. a STABLE radix sort of the stream by term id (hipCUB's device radix sort: a plain library
//      sort, the vendor's primitive) makes it term-major with rows ascending inside a term — the
//      order the search kernel's tile offsets need;
//   4. one pass unpacks rows / tf and writes indptr from the boundaries of the sorted keys.
// All integer work; the result is identical to counting on the host.
#include "common.h"

#include <hipcub/hipcub.hpp>

namespace {

constexpr int BD_NT = 256;             // four documents per workgroup

struct Scratch {                       // device scratch that lives for one build call
    DevBuf b;
    void* ptr() const { return b.p; }
    int reserve(size_t bytes) { return b.ensure(bytes); }
    ~Scratch() { b.release(); }
};
#define BD_TRY(expr) do { const int rc_ = (expr); if (rc_ != RAG_OK) return rc_; } while (0)

struct DocParams {
    const int32_t* term;               // [n_tokens] term ids, document after document; negative = ignored token
    const int64_t* doc_off;            // [n_docs + 1]
    int64_t n_docs;
    int64_t n_tokens;
    int64_t vocab;
    int64_t* n_unique;                 // [n_docs] pass 1 out; pass 2 in: exclusive offsets
    uint32_t* out_term;                // [nnz] pass 2
    uint64_t* out_val;                 // [nnz] (row << 32) | tf
    int64_t capacity;
    uint32_t* fault;
};

template <bool EMIT>
__global__ __launch_bounds__(BD_NT) void doc_unique_kernel(DocParams p) {
    __shared__ int32_t tile[BD_NT / 64][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t d = (int64_t)blockIdx.x * (BD_NT / 64) + wave;
    if (d >= p.n_docs) return;                                 // whole waves leave: no barrier below
    int64_t a = p.doc_off[d], b = p.doc_off[d + 1];
    if (a < 0 || b < a || b > p.n_tokens) {                    // offsets beyond the token array
        if (lane == 0) rag_flag(p.fault, RAG_FAULT_SEGMENT);
        a = 0;
        b = 0;
    }
    const int64_t len = b - a;
    int64_t base = EMIT ? p.n_unique[d] : 0;                   // where this document's postings start
    int64_t uniq = 0;
    for (int64_t i0 = 0; i0 < len; i0 += 64) {
        const int64_t i = i0 + lane;
        int32_t ti = i < len ? p.term[a + i] : -1;
        if (ti >= p.vocab) {                                   // a term id outside the vocabulary
            rag_flag(p.fault, RAG_FAULT_BM25_TERM);
            ti = -1;
        }
        uint32_t tf = 0;
        bool first = ti >= 0;
        for (int64_t j0 = 0; j0 < len; j0 += 64) {
            const int64_t j = j0 + lane;
            tile[wave][lane] = j < len ? p.term[a + j] : -2;
            __builtin_amdgcn_wave_barrier();
            const int n = (int)min((int64_t)64, len - j0);
            for (int k = 0; k < n; ++k) {
                const bool same = tile[wave][k] == ti;
                tf += same ? 1u : 0u;
                first = first && !(same && j0 + k < i);
            }
            __builtin_amdgcn_wave_barrier();
        }
        const unsigned long long firsts = __ballot(first);
        if (EMIT && first) {
            const int64_t at = base + __popcll(firsts & ((1ull << lane) - 1ull));
            if (at < p.capacity) {
                p.out_term[at] = (uint32_t)ti;
                p.out_val[at] = ((uint64_t)d << 32) | tf;
            } else {
                rag_flag(p.fault, RAG_FAULT_BM25_LIST);
            }
        }
        base += __popcll(firsts);
        uniq += __popcll(firsts);
    }
    if (!EMIT && lane == 0) p.n_unique[d] = uniq;
}

// sorted stream -> rows, tf, indptr.  Thread i owns posting i and the indptr entries of the terms
// in (key[i-1], key[i]]; the last thread also those after the last key.
__global__ void unpack_kernel(const uint32_t* key, const uint64_t* val, int64_t nnz, int64_t vocab, int64_t n_docs,
                              uint32_t* rows, uint32_t* tf, int64_t* indptr, uint32_t* fault) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (nnz == 0) {
        for (int64_t t = i; t <= vocab; t += (int64_t)gridDim.x * blockDim.x) indptr[t] = 0;
        return;
    }
    if (i >= nnz) return;
    const uint64_t v = val[i];
    const uint32_t row = (uint32_t)(v >> 32);
    if ((int64_t)row >= n_docs) rag_flag(fault, RAG_FAULT_BM25_POSTING);
    rows[i] = row;
    tf[i] = (uint32_t)v;
    const int64_t k = min((int64_t)key[i], vocab - 1);
    const int64_t prev = i ? min((int64_t)key[i - 1], vocab - 1) : -1;
    for (int64_t t = prev + 1; t <= k; ++t) indptr[t] = i;
    if (i == nnz - 1)
        for (int64_t t = k + 1; t <= vocab; ++t) indptr[t] = nnz;
}

// impact = tf (k1 + 1) / (tf + k1 (1 - b + b len / avgdl)), every operation rounded on its own, in the
// operand order of the upstream expression (k1 + 1 and 1 - b are formed by the caller the same way).
__global__ void impact_kernel(const uint32_t* rows, const uint32_t* tf, int64_t nnz, const int64_t* doc_len, int64_t n_docs,
                              double avgdl, double k1, double b, double k1p1, double one_minus_b, double* impact,
                              uint32_t* fault) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    int64_t row = rows[i];
    if (row >= n_docs) {
        rag_flag(fault, RAG_FAULT_BM25_POSTING);
        row = n_docs - 1;
    }
    const double len = (double)doc_len[row];
    const double norm = __dmul_rn(k1, __dadd_rn(one_minus_b, __ddiv_rn(__dmul_rn(b, len), avgdl)));
    const double f = (double)tf[i];
    impact[i] = __ddiv_rn(__dmul_rn(f, k1p1), __dadd_rn(f, norm));
}

}  // namespace

extern "C" int rag_bm25_build_dev(const int32_t* d_term, int64_t n_tokens, const int64_t* d_doc_off, int64_t n_docs,
                                  int64_t vocab, int64_t capacity, int64_t* d_indptr, uint32_t* d_rows, uint32_t* d_tf,
                                  int64_t* out_nnz, void* stream_) {
    RAG_REQUIRE(out_nnz != nullptr, RAG_EINVAL, "rag_bm25_build: out_nnz is null");
    *out_nnz = 0;
    RAG_REQUIRE(n_docs > 0 && vocab > 0 && n_tokens >= 0 && capacity >= 0, RAG_EINVAL, "rag_bm25_build: bad size");
    RAG_REQUIRE(n_docs < (1ll << 32) && vocab < (1ll << 31) && n_tokens < (1ll << 32), RAG_ELIMIT,
                "rag_bm25_build: rows, vocabulary and tokens must stay below 2^32 / 2^31 / 2^32");
    RAG_REQUIRE(d_doc_off && d_indptr && (n_tokens == 0 || d_term), RAG_EINVAL, "rag_bm25_build: null pointer");
    RAG_REQUIRE(capacity == 0 || (d_rows && d_tf), RAG_EINVAL, "rag_bm25_build: null output");
    hipStream_t stream = (hipStream_t)stream_;
    Scratch uniq, key_in, key_out, val_in, val_out, temp;
    BD_TRY(uniq.reserve((size_t)n_docs * sizeof(int64_t)));
    DocParams p;
    p.term = d_term; p.doc_off = d_doc_off; p.n_docs = n_docs; p.n_tokens = n_tokens; p.vocab = vocab;
    p.n_unique = (int64_t*)uniq.ptr(); p.out_term = nullptr; p.out_val = nullptr; p.capacity = 0;
    p.fault = rag_fault_word();
    const unsigned doc_blocks = (unsigned)ceil_div64(n_docs, BD_NT / 64);
    hipLaunchKernelGGL(doc_unique_kernel<false>, dim3(doc_blocks), dim3(BD_NT), 0, stream, p);
    RAG_HIP_TRY(hipGetLastError());
    // total and exclusive offsets
    int64_t last_count = 0, last_off = 0;
    size_t temp_bytes = 0;
    RAG_HIP_TRY(hipMemcpyAsync(&last_count, p.n_unique + n_docs - 1, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    RAG_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, temp_bytes, p.n_unique, p.n_unique, n_docs, stream));
    BD_TRY(temp.reserve(temp_bytes ? temp_bytes : 16));
    RAG_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(temp.ptr(), temp_bytes, p.n_unique, p.n_unique, n_docs, stream));
    RAG_HIP_TRY(hipMemcpyAsync(&last_off, p.n_unique + n_docs - 1, sizeof(int64_t), hipMemcpyDeviceToHost, stream));
    RAG_HIP_TRY(hipStreamSynchronize(stream));
    const int64_t nnz = last_off + last_count;
    *out_nnz = nnz;
    RAG_REQUIRE(nnz <= capacity, RAG_ELIMIT, "rag_bm25_build: %lld postings, room for %lld", (long long)nnz,
                (long long)capacity);
    RAG_REQUIRE(nnz < (1ll << 32), RAG_ELIMIT, "rag_bm25_build: %lld postings out of range", (long long)nnz);
    const size_t n_alloc = nnz ? (size_t)nnz : 1;
    BD_TRY(key_in.reserve(n_alloc * sizeof(uint32_t)));
    BD_TRY(key_out.reserve(n_alloc * sizeof(uint32_t)));
    BD_TRY(val_in.reserve(n_alloc * sizeof(uint64_t)));
    BD_TRY(val_out.reserve(n_alloc * sizeof(uint64_t)));
    if (nnz) {
        p.out_term = (uint32_t*)key_in.ptr(); p.out_val = (uint64_t*)val_in.ptr(); p.capacity = nnz;
        hipLaunchKernelGGL(doc_unique_kernel<true>, dim3(doc_blocks), dim3(BD_NT), 0, stream, p);
        RAG_HIP_TRY(hipGetLastError());
        int bits = 1;
        while ((1ll << bits) < vocab) ++bits;
        size_t sort_bytes = 0;
        RAG_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const uint32_t*)key_in.ptr(), (uint32_t*)key_out.ptr(),
                                                       (const uint64_t*)val_in.ptr(), (uint64_t*)val_out.ptr(), nnz, 0, bits,
                                                       stream));
        BD_TRY(temp.reserve(sort_bytes ? sort_bytes : 16));
        RAG_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(temp.ptr(), sort_bytes, (const uint32_t*)key_in.ptr(), (uint32_t*)key_out.ptr(),
                                                       (const uint64_t*)val_in.ptr(), (uint64_t*)val_out.ptr(), nnz, 0, bits,
                                                       stream));
    }
    const unsigned blocks = nnz ? (unsigned)ceil_div64(nnz, 256) : (unsigned)ceil_div64(vocab + 1, 256);
    hipLaunchKernelGGL(unpack_kernel, dim3(blocks), dim3(256), 0, stream, (const uint32_t*)key_out.ptr(),
                       (const uint64_t*)val_out.ptr(), nnz, vocab, n_docs, d_rows, d_tf, d_indptr, p.fault);
    RAG_HIP_TRY(hipGetLastError());
    RAG_HIP_TRY(hipStreamSynchronize(stream));          // the temporaries are released on return
    return RAG_OK;
}

extern "C" int rag_bm25_impacts_dev(const uint32_t* d_rows, const uint32_t* d_tf, int64_t nnz, const int64_t* d_doc_len,
                                    int64_t n_docs, double avgdl, double k1, double b, double* d_impact, void* stream_) {
    RAG_REQUIRE(nnz >= 0 && n_docs > 0, RAG_EINVAL, "rag_bm25_impacts: bad size");
    if (nnz == 0) return RAG_OK;
    RAG_REQUIRE(d_rows && d_tf && d_doc_len && d_impact, RAG_EINVAL, "rag_bm25_impacts: null pointer");
    RAG_REQUIRE(avgdl > 0.0, RAG_EINVAL, "rag_bm25_impacts: average document length must be positive");
    hipLaunchKernelGGL(impact_kernel, dim3((unsigned)ceil_div64(nnz, 256)), dim3(256), 0, (hipStream_t)stream_, d_rows, d_tf,
                       nnz, d_doc_len, n_docs, avgdl, k1, b, k1 + 1.0, 1.0 - b, d_impact, rag_fault_word());
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}
