// Shared host/device helpers for librag_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>

#include "../../include/rag_hip.h"

// ---------------------------------------------------------------------------------
// host side: error reporting
// ---------------------------------------------------------------------------------
void rag_set_error(const char* fmt, ...);

#define RAG_HIP_TRY(expr)                                                              \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            rag_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                          __FILE__, __LINE__);                                         \
            return RAG_EHIP;                                                           \
        }                                                                              \
    } while (0)

#define RAG_REQUIRE(cond, code, ...)                                                   \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            rag_set_error(__VA_ARGS__);                                                \
            return (code);                                                             \
        }                                                                              \
    } while (0)

// Grow-only device buffer owned by a handle (steady-state calls never allocate).
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return RAG_OK;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            cap = 0;
            if (e != hipSuccess) { rag_set_error("hipFree failed: %s", hipGetErrorString(e)); return RAG_EHIP; }
        }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            rag_set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            return RAG_ENOMEM;
        }
        cap = want;
        return RAG_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Raise a kernel's dynamic-LDS limit to `bytes` on the CURRENT device (the attribute belongs to the
// device's copy of the function, so the grants are kept per device and kernel; a process may drive
// several GPUs).  Cheap when already granted.  Returns RAG_OK or RAG_EHIP (error text set).
int rag_grant_dyn_lds(const void* kernel, size_t bytes);
#define RAG_GRANT_LDS(kernel, bytes)                                                   \
    do {                                                                               \
        int _rc = rag_grant_dyn_lds((const void*)(kernel), (size_t)(bytes));            \
        if (_rc != RAG_OK) return _rc;                                                 \
    } while (0)

// Device word of the current device that collects RAG_FAULT_* bits (null if it could not be
// allocated).  Kernels receive it as an argument; a kernel that finds an index read from
// device data outside its array sets a bit there and carries on with a safe value instead
// of making the access.
uint32_t* rag_fault_word();
bool rag_fault_word_ready();

// a stream under graph capture records no timing events (an event recorded into a graph cannot be read back)
static inline bool rag_stream_capturing(hipStream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
}

// Once per device, with that device current: look up the device address of the static work array behind
// rag_bm25_search_static_dev (bm25.hip).  Called by rag_init.
int rag_bm25_static_setup();

__device__ __forceinline__ void rag_flag(uint32_t* w, uint32_t bit) {
    if (w != nullptr) atomicOr(w, bit);
}

// ---------------------------------------------------------------------------------
// device side: order-preserving integer keys for floating-point scores
// (larger key <=> larger score; -0 and +0 map to the same key)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f32_to_key(float s) {
    s += 0.0f;                               // -0 -> +0
    uint32_t u = __float_as_uint(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_f32(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t f64_to_key(double s) {
    s += 0.0;
    uint64_t u = (uint64_t)__double_as_longlong(s);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_to_f64(uint64_t k) {
    uint64_t u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

// ---------------------------------------------------------------------------------
// device side: exact top-K threshold by most-significant-digit radix select
//
// Items are (key, row) pairs ordered lexicographically, larger = better; rows are
// unique, so the K-th best item is unique.  radix_select returns a masked-prefix
// threshold `t` such that exactly K items satisfy sel_pass(t, key, row).
// ---------------------------------------------------------------------------------
template <typename KeyT>
struct SelThr {
    KeyT pk, mk;       // determined high bytes of the key and their mask
    uint32_t pr, mr;   // determined high bytes of the row and their mask
};

template <typename KeyT>
__device__ __forceinline__ bool sel_pass(const SelThr<KeyT>& t, KeyT key, uint32_t row) {
    KeyT a = key & t.mk;
    return a > t.pk || (a == t.pk && (row & t.mr) >= t.pr);
}

// LDS scratch needed by one radix_select group: 256 histogram bins + 8 broadcast words,
// 16-byte aligned.
#define RAG_SEL_LDS_WORDS 264

// NT = threads in the cooperating group: 64 (one wave, other waves of the block may be
// doing something else) or the whole block (then every thread of the block must call).
// get(i, key, row) loads item i; n = number of items; requires 1 <= K <= n.
//
// Keys of one list usually share their high bits (scores of one query lie in a narrow
// range), so the key passes start at the highest bit in which any two keys differ: the first
// histogram then spreads the items over its 256 bins instead of piling them onto one.
template <typename KeyT, int NT, typename Get>
__device__ inline SelThr<KeyT> radix_select(Get get, int n, int K, uint32_t* lds, int tid) {
    uint32_t* hist = lds;
    uint32_t* bc = lds + 256;          // [0..2] digit/above/inbin, [4..5] OR of key differences
    SelThr<KeyT> t;
    t.pk = 0; t.mk = 0; t.pr = 0; t.mr = 0;
    uint32_t krem = (uint32_t)K;

    auto gsync = [&]() {
        if constexpr (NT == 64) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            __syncthreads();
        }
    };
    // one wave finds the digit d with  #(digit > d) < krem <= #(digit >= d)
    auto pick_digit = [&]() {
        if (tid < 64) {
            const int l = tid;
            uint32_t h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
            uint32_t s = h0 + h1 + h2 + h3;
            uint32_t v = s;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t o = __shfl_down(v, off, 64);
                if (l + off < 64) v += o;
            }
            uint32_t c = v - s;                    // items in bins above this lane's four
            uint32_t hh[4] = {h0, h1, h2, h3};
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                if (c < krem && krem <= c + hh[j]) {
                    // digit, items strictly above it, items in it.  One aligned 16-byte store on
                    // purpose: written as three dwords the compiler merges them into one
                    // ds_write_b96, and with that store selections came out a few items short
                    // (histogram totals of consecutive passes disagreed) whenever a second
                    // workgroup shared the compute unit (gfx950, ROCm 7.2); with ds_write_b128 or
                    // three ds_write_b32 they never did.
                    *reinterpret_cast<uint4*>(bc) = make_uint4((uint32_t)(4 * l + j), c, hh[j], 1u);
                }
                c += hh[j];
            }
        }
    };

    // ---- which key bits differ at all?
    if (tid == 0) { bc[4] = 0; bc[5] = 0; }
    gsync();
    {
        KeyT key0; uint32_t row0;
        get(0, key0, row0);
        KeyT diff = 0;
        for (int i = tid; i < n; i += NT) {
            KeyT key; uint32_t row;
            get(i, key, row);
            diff |= key ^ key0;
        }
        if (diff) {
            atomicOr(&bc[4], (uint32_t)diff);
            if constexpr (sizeof(KeyT) == 8) atomicOr(&bc[5], (uint32_t)((uint64_t)diff >> 32));
        }
        gsync();
        KeyT all = (KeyT)bc[4];
        if constexpr (sizeof(KeyT) == 8) all |= (KeyT)((uint64_t)bc[5] << 32);
        // smear the highest differing bit downwards: low = every bit position at or below it
        KeyT low = all;
        low |= low >> 1; low |= low >> 2; low |= low >> 4; low |= low >> 8; low |= low >> 16;
        if constexpr (sizeof(KeyT) == 8) low |= low >> 32;
        t.mk = ~low;                                        // the keys agree on these bits
        t.pk = key0 & t.mk;
        const int n_low = sizeof(KeyT) == 8 ? __popcll((unsigned long long)low) : __popc((unsigned int)low);
        bool done = false;
        // ---- key passes, 8 bits at a time from bit `hi` down
        int shift = n_low;
        while (shift > 0 && !done) {
            const int lo = shift > 8 ? shift - 8 : 0;
            const uint32_t wmask = (1u << (shift - lo)) - 1u;
            for (int b = tid; b < 256; b += NT) hist[b] = 0;
            gsync();
            for (int i = tid; i < n; i += NT) {
                KeyT key; uint32_t row;
                get(i, key, row);
                if ((key & t.mk) == t.pk) atomicAdd(&hist[(uint32_t)(key >> lo) & wmask], 1u);
            }
            gsync();
            pick_digit();
            gsync();
            const uint4 pick = *reinterpret_cast<const uint4*>(bc);
            const uint32_t d = pick.x, above = pick.y, inbin = pick.z;
            t.pk |= (KeyT)d << lo;
            t.mk |= (KeyT)wmask << lo;
            krem -= above;
            done = inbin == krem;                       // every item of this bin is wanted
            shift = lo;
        }
        // ---- equal keys: order by row, one byte at a time
        for (int pass = 0; pass < 4 && !done; ++pass) {
            const int lo = 8 * (3 - pass);
            for (int b = tid; b < 256; b += NT) hist[b] = 0;
            gsync();
            for (int i = tid; i < n; i += NT) {
                KeyT key; uint32_t row;
                get(i, key, row);
                if (key == t.pk && (row & t.mr) == t.pr) atomicAdd(&hist[(row >> lo) & 255u], 1u);
            }
            gsync();
            pick_digit();
            gsync();
            const uint4 pick = *reinterpret_cast<const uint4*>(bc);
            const uint32_t d = pick.x, above = pick.y, inbin = pick.z;
            t.pr |= d << lo;
            t.mr |= 255u << lo;
            krem -= above;
            done = inbin == krem;
        }
    }
    gsync();
    return t;
}
