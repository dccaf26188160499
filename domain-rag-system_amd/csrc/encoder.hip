// Transformer-encoder kernels for the two BERT forwards of the path (bi-encoder behind
// MedicalEmbedder.encode, reference src/indexing/embedder.py:65-71; cross-encoder behind
// CrossEncoderReranker.score_pairs, reference src/reranker/rerank.py:79-84).  The GEMMs stay
// with hipBLASLt; these kernels replace what sits between them.  gfx950 only.
//
//  attention_kernel   softmax(Q K^T / sqrt(D) + key mask) V for one (sequence, head) per
//                     workgroup, straight from the packed [tokens][3][heads][D] projection
//                     output to the [tokens][heads*D] layout the output projection reads
//                     (no transposes, no score matrix in memory).  bf16 or fp16 in/out, fp32 softmax,
//                     v_mfma_f32_32x32x16_bf16 / _f16 for both products.
//  add_layernorm_kernel   LayerNorm(x + residual) in one pass, fp32 arithmetic, one wave per row.
#include "common.h"

#include <algorithm>

#include <math.h>

namespace {

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16;

// The 16-bit type a split-precision operand is held in: x = hi + lo with hi = H(x), lo = H(x - hi).
// bf16 halves carry 16 significant bits between them (products good to 2^-16), fp16 halves 22 (products as
// good as an fp32 product; fp16 subnormals are kept by the matrix cores, so small values lose nothing that
// matters: absolute resolution 2^-25).  Both run at the same matrix-core rate.
template <typename H> struct Half16;
template <> struct Half16<__bf16> {
    typedef __bf16 x8 __attribute__((ext_vector_type(8)));
    typedef __bf16 x4 __attribute__((ext_vector_type(4)));
    typedef __bf16 x2 __attribute__((ext_vector_type(2)));
    __device__ static f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Half16<_Float16> {
    typedef _Float16 x8 __attribute__((ext_vector_type(8)));
    typedef _Float16 x4 __attribute__((ext_vector_type(4)));
    typedef _Float16 x2 __attribute__((ext_vector_type(2)));
    __device__ static f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// x = hi + lo with hi = H(x), lo = H(x - hi).  The hi that is stored has to be the hi that is subtracted: left to
// itself the compiler folds a multiply or add that produced x into ONE of the two conversions (v_fma_mixlo_f16
// rounds the exact product straight to fp16, the other path rounds to fp32 first), and next to a rounding tie
// the two disagree by one fp16 ulp: hi + lo is then off by that ulp (seen as errors of 2^-10 in one element in
// ten thousand).  Passing x through an opaque register makes it a finished fp32 value for both.
template <typename H16>
__device__ __forceinline__ float finished_f32(float x) {
    asm volatile("" : "+v"(x));
    return x;
}
#define RAG_SPLIT_HI_LO(H16, x, hi, lo)                  \
    do {                                                 \
        const float _x = finished_f32<H16>(x);           \
        (hi) = (H16)_x;                                  \
        (lo) = (H16)(_x - (float)(hi));                  \
    } while (0)

constexpr int ATT_NT = 256;          // at most 4 waves; each takes 32-query blocks round-robin
constexpr float NEG_BIG = -1e30f;    // finite stand-in for -inf (keeps max/exp free of NaN)

// 8-element vector access for the row kernels
template <typename T> struct Vec8;
template <> struct Vec8<bf16> {
    typedef bf16x8 type;
    __device__ static void load(const bf16* p, float* f) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (float)v[j];
    }
    __device__ static void store(bf16* p, const float* f) {
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (bf16)f[j];
        *reinterpret_cast<bf16x8*>(p) = v;
    }
};
template <> struct Vec8<f16> {
    typedef Half16<f16>::x8 type;
    __device__ static void load(const f16* p, float* f) {
        const type v = *reinterpret_cast<const type*>(p);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (float)v[j];
    }
    __device__ static void store(f16* p, const float* f) {
        type v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (f16)f[j];
        *reinterpret_cast<type*>(p) = v;
    }
};
template <> struct Vec8<float> {
    __device__ static void load(const float* p, float* f) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
    __device__ static void store(float* p, const float* f) {
        *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
};

constexpr int LN_MAX_CHUNKS = 4;     // 8-element chunks per lane: hidden sizes up to 2048

// The two lanes of a query (32 apart) combine or trade registers with one cross-half swap instruction
// (v_permlane32_swap: the upper half of the first operand changes places with the lower half of the second) instead
// of a round trip through the LDS crossbar.
__device__ __forceinline__ float both_halves_max(float v) {
    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
}
__device__ __forceinline__ float both_halves_sum(float v) {
    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
}
// afterwards the lower lane holds (its a, the upper lane's a) and the upper lane (the lower lane's b, its b)
__device__ __forceinline__ void trade_halves(uint32_t& a, uint32_t& b) {
    const auto r2 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r2[0];
    b = r2[1];
}
__device__ __forceinline__ void trade_halves(float& a, float& b) {
    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r2[0]);
    b = __uint_as_float(r2[1]);
}

// LDS image of one head's keys and values:
//   K   [kpad][D + 8]     rows padded by 16 B: the 16-byte operand reads of 32 keys spread over the banks
//   V^T [D][kpad + 4]     rows padded by 8 B: 32 lanes read 8 B of 32 different d rows without conflict
#ifndef RAG_ATT_MIN_BLOCKS
#define RAG_ATT_MIN_BLOCKS 6         // workgroups per CU the register budget is sized for
#endif
template <int D, typename E>
__global__ __launch_bounds__(ATT_NT, RAG_ATT_MIN_BLOCKS) void attention_kernel(const E* __restrict__ qkv,
                                                           const int32_t* __restrict__ lens,
                                                           const int32_t* __restrict__ cu, int64_t total_tokens,
                                                           int n_seq, int S_max, int NH, float scale,
                                                           int first_only, E* __restrict__ out, uint32_t* fault) {
    // first_only: only the first token of every sequence is a query ([CLS] of the last layer, the
    // one row the pooler reads); out is then [n_seq][H]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Half16<E>::x8 e8;
    typedef typename Half16<E>::x4 e4;
    typedef typename Half16<E>::x2 e2;
    constexpr int KROW = D + 8;
    constexpr int DT = D / 32;                    // 32-row tiles of the transposed output
    constexpr int KC = D / 16;                    // k-steps of the score product
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // The heads of one sequence interleave in memory (a head's 64- or 128-byte slice of a token
    // shares its cache lines with the neighbouring heads), so all heads of a sequence go to
    // workgroups with the same blockIdx % 8, which round-robin placement puts on one XCD: the
    // lines are then fetched into one L2 once instead of into several.
    const int jb = blockIdx.x >> 3;
    const int head = jb % NH;
    const int seq = (jb / NH) * 8 + (blockIdx.x & 7);
    if (seq >= n_seq) return;
    const int H = NH * D;
    const size_t tok = (size_t)3 * H;             // elements per token of the projection output
    // padded batch: sequence i owns rows [i*S_max, (i+1)*S_max), lens[i] of them real;
    // packed batch: sequence i owns rows [cu[i], cu[i+1]), all real
    int64_t first;
    int S, len;
    if (cu != nullptr) {
        first = cu[seq];
        const int64_t n = (int64_t)cu[seq + 1] - first;
        if (first < 0 || n < 0 || n > S_max || first + n > total_tokens) {   // not a sequence of this batch
            if (threadIdx.x == 0) rag_flag(fault, RAG_FAULT_SEGMENT);
            return;
        }
        len = (int)n;
        S = len;
    } else {
        first = (int64_t)seq * S_max;
        S = S_max;
        len = min(max(lens[seq], 0), S);
    }
    const E* base = qkv + (size_t)first * tok;
    E* obase = first_only ? out + (size_t)seq * H + head * D : out + (size_t)first * H + head * D;
    const int kpad = (len + 31) & ~31;
    const int VROW = kpad + 4;
    E* Ks = reinterpret_cast<E*>(smem);
    E* Vt = Ks + (size_t)kpad * KROW;

    // ---- stage this head's keys (row-major) and values (transposed), zero beyond len
    // Q^T fragments (B operand) of this wave's first query block: lane (h, r) holds Q[q][16c + 8h + j].
    // Issued before the K/V staging so that both round trips to memory overlap.
    const int n_query_rows = first_only ? 1 : S;
    e8 qf[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[c][j] = (E)0.f;
        if (wave * 32 + r < S && wave * 32 < min(n_query_rows, len))
            qf[c] = *reinterpret_cast<const e8*>(base + (size_t)(wave * 32 + r) * tok + head * D + 16 * c + 8 * h);
    }
    // each thread moves 8 dims of two neighbouring keys, so the transposed image is written 4 bytes at a time
    for (int idx = tid; idx < (kpad / 2) * (D / 8); idx += (int)blockDim.x) {
        const int key = (idx / (D / 8)) * 2, c8 = idx % (D / 8);
        e8 k0, k1, v0, v1;
#pragma unroll
        for (int j = 0; j < 8; ++j) { k0[j] = (E)0.f; k1[j] = (E)0.f; v0[j] = (E)0.f; v1[j] = (E)0.f; }
        const E* p = base + (size_t)key * tok + head * D + c8 * 8;
        if (key < len) {
            k0 = *reinterpret_cast<const e8*>(p + H);
            v0 = *reinterpret_cast<const e8*>(p + 2 * H);
        }
        if (key + 1 < len) {
            k1 = *reinterpret_cast<const e8*>(p + tok + H);
            v1 = *reinterpret_cast<const e8*>(p + tok + 2 * H);
        }
        *reinterpret_cast<e8*>(Ks + key * KROW + c8 * 8) = k0;
        *reinterpret_cast<e8*>(Ks + (key + 1) * KROW + c8 * 8) = k1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            e2 pr;
            pr[0] = v0[j];
            pr[1] = v1[j];
            *reinterpret_cast<e2*>(Vt + (c8 * 8 + j) * VROW + key) = pr;
        }
    }
    __syncthreads();

    for (int qb = wave; qb * 32 < n_query_rows; qb += (int)(blockDim.x >> 6)) {
        const int q = qb * 32 + r;
        if (qb * 32 >= len) {
            // padding rows only: defined (zero) output, nothing to compute
            for (int i = lane; i < 32 * (D / 8); i += 64) {
                const int row = qb * 32 + i / (D / 8), c8 = i % (D / 8);
                if (row < n_query_rows) {
                    e8 z;
#pragma unroll
                    for (int j = 0; j < 8; ++j) z[j] = (E)0.f;
                    *reinterpret_cast<e8*>(obase + (size_t)row * H + c8 * 8) = z;
                }
            }
            continue;
        }
        if (qb != wave) {                                // later blocks of a long sequence
#pragma unroll
            for (int c = 0; c < KC; ++c) {
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[c][j] = (E)0.f;
                if (q < S) qf[c] = *reinterpret_cast<const e8*>(base + (size_t)q * tok + head * D + 16 * c + 8 * h);
            }
        }
        float m_run = NEG_BIG, l_run = 0.f;
        f32x16 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;

        for (int kb = 0; kb < kpad; kb += 32) {
            // S^T tile: rows = keys kb..kb+31 (registers + lane half), column = this lane's query
            f32x16 s;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const e8 a = *reinterpret_cast<const e8*>(Ks + (kb + r) * KROW + 16 * c + 8 * h);
                s = Half16<E>::mfma(a, qf[c], s);
            }
            // softmax in base 2: t = s * (scale * log2 e), p = 2^(t - m); only the last key block
            // can hold keys beyond len
            float mx = NEG_BIG;
            if (kb + 32 > len) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kb + (i & 3) + 8 * (i >> 2) + 4 * h;
                    s[i] = key < len ? s[i] * scale : NEG_BIG;
                    mx = fmaxf(mx, s[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    s[i] *= scale;
                    mx = fmaxf(mx, s[i]);
                }
            }
            mx = both_halves_max(mx);
            const float m_new = fmaxf(m_run, mx);
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s[i] = __builtin_amdgcn_exp2f(s[i] - m_new);      // masked keys: 2^(-1e30 - m) = 0
                sum += s[i];
            }
            sum = both_halves_sum(sum);
            if (kb > 0) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run = l_run * alpha + sum;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
            } else {
                l_run = sum;
            }
            m_run = m_new;
            // O^T += V^T P^T: the score registers 8t..8t+7 are the B fragment of k-step t; element j of
            // lane half h is key kb + 16t + 8(j>>2) + 4h + (j&3), so V^T is read in that order
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                e8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[j] = (E)s[8 * t + j];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const E* vp = Vt + (size_t)(dt * 32 + r) * VROW + kb + 16 * t + 4 * h;
                    const e4 lo = *reinterpret_cast<const e4*>(vp);
                    const e4 hi = *reinterpret_cast<const e4*>(vp + 8);
                    e8 av;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { av[j] = lo[j]; av[4 + j] = hi[j]; }
                    o[dt] = Half16<E>::mfma(av, pb, o[dt]);
                }
            }
        }
        // O^T tile: rows = d (registers + lane half), column = this lane's query
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        // A lane holds d = 8g + 4h + j of its query's row (g = 0..3): the two lanes of a query swap half of
        // their groups, after which the lower one owns d 0..7 and 16..23, the upper one d 8..15 and 24..31,
        // and each store instruction writes 32 contiguous bytes per row in 16-byte pieces (the 8-byte pieces
        // of the direct form cost 13 % of the kernel's time).
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            uint2 grp[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                e4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (E)(o[dt][4 * g + j] * inv);
                grp[g] = __builtin_bit_cast(uint2, v);
            }
            // lower lane sends groups 1, 3 and keeps 0, 2; upper lane sends 0, 2 and keeps 1, 3
            // 8 elements = (h = 0 part, h = 1 part) of one group: lower g0 then g2, upper g1 then g3
            trade_halves(grp[0].x, grp[1].x); trade_halves(grp[0].y, grp[1].y);
            trade_halves(grp[2].x, grp[3].x); trade_halves(grp[2].y, grp[3].y);
            const uint2 a0 = grp[0], a1 = grp[1], b0 = grp[2], b1 = grp[3];
            if (q < n_query_rows) {
                E* dst = obase + (size_t)q * H + dt * 32 + 8 * h;
                *reinterpret_cast<uint4*>(dst) = make_uint4(a0.x, a0.y, a1.x, a1.y);
                *reinterpret_cast<uint4*>(dst + 16) = make_uint4(b0.x, b0.y, b1.x, b1.y);
            }
        }
    }
}

// fp32-equivalent attention on the bf16 matrix cores, for the float32 cross-encoder forward (the
// reference's CrossEncoder.predict runs in float32, src/reranker/rerank.py:79-84).  Same structure as
// attention_kernel (one workgroup per (sequence, head), keys row-major and values transposed in LDS,
// S^T = K Q^T with a query per lane, probabilities fed back as the B operand of O^T += V^T P^T), with
// every operand split into bf16 high and low parts, x = hi + lo, and every product taken as
// hi.hi + lo.hi + hi.lo with fp32 accumulation (the dropped lo.lo term is below 2^-17 of the
// product).  Input is the fp32 q|k|v projection WITHOUT its bias (added here while staging, so the
// GEMM output needs no bias pass); output is the attention result already split for the output
// projection's GEMM: row = [hi | hi | lo], 3 * H bf16 (the left operand of [w_hi ; w_lo ; w_hi]).
// Packed batches only (cu_seqlens).
template <int D, typename H16>
__global__ __launch_bounds__(ATT_NT, 3) void attention_split_kernel(const float* __restrict__ qkv,
                                                                    const float* __restrict__ bias,
                                                                    const int32_t* __restrict__ cu, int64_t total_tokens,
                                                                    int n_seq, int S_max, int NH, float scale,
                                                                    int first_only, int out_hl, H16* __restrict__ out,
                                                                    uint32_t* fault) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename Half16<H16>::x8 hx8;
    typedef typename Half16<H16>::x4 hx4;
    typedef typename Half16<H16>::x2 hx2;
    // Head size 32: both images in 64-byte rows [key][32] with nothing between them (256 bytes of LDS per key for the four
    // images: a workgroup more per CU than padded rows allow).  Keys: the 16-byte piece p of row `key` sits at
    // p ^ ((key >> 2) & 3), so that 16 consecutive rows read at one piece index cover all 64 banks once.  Values stay
    // row-major too and the V^T operand comes out of that image through the transposing LDS read (ds_read_b64_tr_b16:
    // a 16-lane group reads a block of 4 keys x 16 dimensions, each lane receives one dimension's 4 keys; with 64-byte
    // rows the four groups' blocks tile the 64 banks exactly), so staging writes 16-byte pieces instead of transposing
    // through 4-byte ones.  Head size 64: keys in rows padded by 8, values transposed [dimension][key] in rows of kpad + 4.
    constexpr bool TR = D == 32;
    constexpr int KROW = TR ? D : D + 8;
    constexpr int DT = D / 32;
    constexpr int KC = D / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int jb = blockIdx.x >> 3;
    const int head = jb % NH;
    const int seq = (jb / NH) * 8 + (blockIdx.x & 7);    // a sequence's heads share an XCD (see attention_kernel)
    if (seq >= n_seq) return;
    const int H = NH * D;
    const size_t tok = (size_t)3 * H;
    const int64_t first = cu[seq];
    const int64_t n = (int64_t)cu[seq + 1] - first;
    if (first < 0 || n < 0 || n > S_max || first + n > total_tokens) {
        if (threadIdx.x == 0) rag_flag(fault, RAG_FAULT_SEGMENT);
        return;
    }
    const int len = (int)n;
    const float* base = qkv + (size_t)first * tok + head * D;
    const float* bq = bias != nullptr ? bias + head * D : nullptr;
    const size_t otok = out_hl ? (size_t)2 * H : tok;     // output row: [hi | lo] or [hi | hi | lo]
    H16* obase = (first_only ? out + (size_t)seq * otok : out + (size_t)first * otok) + head * D;
    const int kpad = (len + 31) & ~31;
    const int VROW = TR ? D : kpad + 4;
    H16* Kh = reinterpret_cast<H16*>(smem);
    H16* Kl = Kh + (size_t)kpad * KROW;
    H16* Vh = Kl + (size_t)kpad * KROW;
    H16* Vl = Vh + (TR ? (size_t)kpad * D : (size_t)D * VROW);
    const int n_query_rows = first_only ? min(1, len) : len;

    // eight consecutive fp32 values (+ bias) -> bf16 high and low parts
    auto load8 = [&](const float* p, const float* b, bool live, hx8& hi, hx8& lo) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (live) {
            const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
            if (b != nullptr) {
                const float4 ba = *reinterpret_cast<const float4*>(b), bc = *reinterpret_cast<const float4*>(b + 4);
                v[0] += ba.x; v[1] += ba.y; v[2] += ba.z; v[3] += ba.w; v[4] += bc.x; v[5] += bc.y; v[6] += bc.z; v[7] += bc.w;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            RAG_SPLIT_HI_LO(H16, v[j], hi[j], lo[j]);
        }
    };
    auto load_q = [&](int q, hx8* qh, hx8* ql) {
#pragma unroll
        for (int c = 0; c < KC; ++c)
            load8(base + (size_t)q * tok + 16 * c + 8 * h, bq != nullptr ? bq + 16 * c + 8 * h : nullptr,
                  q < n_query_rows, qh[c], ql[c]);
    };
    hx8 qh[KC], ql[KC];
    load_q(wave * 32 + r, qh, ql);

    for (int idx = tid; idx < (kpad / 2) * (D / 8); idx += (int)blockDim.x) {
        const int key = (idx / (D / 8)) * 2, c8 = idx % (D / 8);
        const float* p = base + (size_t)key * tok + c8 * 8;
        const float* bk = bq != nullptr ? bq + H + c8 * 8 : nullptr;
        const float* bv = bq != nullptr ? bq + 2 * H + c8 * 8 : nullptr;
        hx8 k0h, k0l, k1h, k1l, v0h, v0l, v1h, v1l;
        load8(p + H, bk, key < len, k0h, k0l);
        load8(p + 2 * H, bv, key < len, v0h, v0l);
        load8(p + tok + H, bk, key + 1 < len, k1h, k1l);
        load8(p + tok + 2 * H, bv, key + 1 < len, v1h, v1l);
        const int kc = TR ? (c8 ^ ((key >> 2) & 3)) * 8 : c8 * 8;       // (key and key + 1 share key >> 2: key is even)
        *reinterpret_cast<hx8*>(Kh + key * KROW + kc) = k0h;
        *reinterpret_cast<hx8*>(Kl + key * KROW + kc) = k0l;
        *reinterpret_cast<hx8*>(Kh + (key + 1) * KROW + kc) = k1h;
        *reinterpret_cast<hx8*>(Kl + (key + 1) * KROW + kc) = k1l;
        if constexpr (TR) {
            *reinterpret_cast<hx8*>(Vh + key * D + c8 * 8) = v0h;
            *reinterpret_cast<hx8*>(Vl + key * D + c8 * 8) = v0l;
            *reinterpret_cast<hx8*>(Vh + (key + 1) * D + c8 * 8) = v1h;
            *reinterpret_cast<hx8*>(Vl + (key + 1) * D + c8 * 8) = v1l;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                hx2 ph, pl;
                ph[0] = v0h[j]; ph[1] = v1h[j];
                pl[0] = v0l[j]; pl[1] = v1l[j];
                *reinterpret_cast<hx2*>(Vh + (c8 * 8 + j) * VROW + key) = ph;
                *reinterpret_cast<hx2*>(Vl + (c8 * 8 + j) * VROW + key) = pl;
            }
        }
    }
    __syncthreads();
    // transposing read: lane 4q + p of a 16-lane group g addresses row q, dimensions 4p.. of its group's block and
    // receives dimension (lane % 16) of rows 0..3: with blocks at dimensions 16 (g & 1) and keys 4 (g >> 1) that is the
    // operand layout of the other form (dimension r = lane % 32, keys 4h .. 4h + 3)
    typedef short tr4 __attribute__((__vector_size__(4 * sizeof(short))));
    const int tr_off = ((lane & 15) >> 2) * D + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    auto tr_read = [](const H16* p) {
        return __builtin_bit_cast(hx4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4*)(p)));
    };
    const int k_swz = TR ? (r >> 2) & 3 : 0;                  // (kb is a multiple of 32: (kb + r) >> 2 and r >> 2 agree mod 4)

    for (int qb = wave; qb * 32 < n_query_rows; qb += (int)(blockDim.x >> 6)) {
        const int q = qb * 32 + r;
        if (qb != wave) load_q(q, qh, ql);
        float m_run = NEG_BIG, l_run = 0.f;
        f32x16 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;

        for (int kb = 0; kb < kpad; kb += 32) {
            f32x16 s;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const int kc = TR ? ((2 * c + h) ^ k_swz) * 8 : 16 * c + 8 * h;
                const hx8 ah = *reinterpret_cast<const hx8*>(Kh + (kb + r) * KROW + kc);
                const hx8 al = *reinterpret_cast<const hx8*>(Kl + (kb + r) * KROW + kc);
                s = Half16<H16>::mfma(al, qh[c], s);
                s = Half16<H16>::mfma(ah, ql[c], s);
                s = Half16<H16>::mfma(ah, qh[c], s);
            }
            float mx = NEG_BIG;
            if (kb + 32 > len) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kb + (i & 3) + 8 * (i >> 2) + 4 * h;
                    s[i] = key < len ? s[i] * scale : NEG_BIG;
                    mx = fmaxf(mx, s[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    s[i] *= scale;
                    mx = fmaxf(mx, s[i]);
                }
            }
            mx = both_halves_max(mx);
            const float m_new = fmaxf(m_run, mx);
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                // the bare v_exp_f32: arguments are <= 0, and below -126 the result is as good as zero either way
                // (exp2f() wraps every call in a five-instruction range extension, a third of this loop's vector work)
                s[i] = __builtin_amdgcn_exp2f(s[i] - m_new);
                sum += s[i];
            }
            sum = both_halves_sum(sum);
            if (kb > 0) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run = l_run * alpha + sum;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
            } else {
                l_run = sum;
            }
            m_run = m_new;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                hx8 pbh, pbl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    RAG_SPLIT_HI_LO(H16, s[8 * t + j], pbh[j], pbl[j]);
                }
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    hx4 h0, h1, l0, l1;
                    if constexpr (TR) {
                        const int vo = (kb + 16 * t + 4 * h) * D + tr_off;
                        h0 = tr_read(Vh + vo); h1 = tr_read(Vh + vo + 8 * D);
                        l0 = tr_read(Vl + vo); l1 = tr_read(Vl + vo + 8 * D);
                    } else {
                        const size_t vo = (size_t)(dt * 32 + r) * VROW + kb + 16 * t + 4 * h;
                        h0 = *reinterpret_cast<const hx4*>(Vh + vo); h1 = *reinterpret_cast<const hx4*>(Vh + vo + 8);
                        l0 = *reinterpret_cast<const hx4*>(Vl + vo); l1 = *reinterpret_cast<const hx4*>(Vl + vo + 8);
                    }
                    hx8 avh, avl;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { avh[j] = h0[j]; avh[4 + j] = h1[j]; avl[j] = l0[j]; avl[4 + j] = l1[j]; }
                    o[dt] = Half16<H16>::mfma(avl, pbh, o[dt]);
                    o[dt] = Half16<H16>::mfma(avh, pbl, o[dt]);
                    o[dt] = Half16<H16>::mfma(avh, pbh, o[dt]);
                }
            }
        }
        // a lane holds d = 8g + 4h + j of its query's row; the two lanes of a query swap half of their groups so
        // that each owns runs of eight consecutive d (lower: 0..7, 16..23; upper: 8..15, 24..31): 16-byte stores
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            float grp[4][4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) grp[g][j] = o[dt][4 * g + j] * inv;
            float run[2][8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                trade_halves(grp[0][j], grp[1][j]);
                trade_halves(grp[2][j], grp[3][j]);
                run[0][j] = grp[0][j];
                run[0][4 + j] = grp[1][j];
                run[1][j] = grp[2][j];
                run[1][4 + j] = grp[3][j];
            }
            if (q < n_query_rows) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    hx8 hi, lo;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        RAG_SPLIT_HI_LO(H16, run[u][j], hi[j], lo[j]);
                    }
                    H16* dst = obase + (size_t)q * otok + dt * 32 + 8 * h + 16 * u;
                    *reinterpret_cast<hx8*>(dst) = hi;
                    if (out_hl) {
                        *reinterpret_cast<hx8*>(dst + H) = lo;
                    } else {
                        *reinterpret_cast<hx8*>(dst + H) = hi;
                        *reinterpret_cast<hx8*>(dst + 2 * H) = lo;
                    }
                }
            }
        }
    }
}

// fp32 attention for the bi-encoder (its embeddings must hold 1e-4 on cosine scores, so its
// forward stays in fp32): one wave per (sequence, head).  Queries are short (a dozen tokens), so a lane
// per query row would leave most of the wave idle: the wave takes 16 query rows at a time and the four
// lanes of a row own a quarter of the head's dimensions each (partial dot products, summed across the
// four with two shuffles; a quarter of the output row each).  Keys and values go through LDS in blocks
// of 16 (8 KB for 64 dimensions, so a dozen waves fit a compute unit), loaded with every request of a
// block in flight at once; two-pass softmax per block with a running maximum.
// The arithmetic is tiny next to the GEMMs (a few GFLOP per batch); plain FMAs, no MFMA.
template <int D>
__global__ __launch_bounds__(64) void attention_f32_kernel(const float* __restrict__ qkv,
                                                           const int32_t* __restrict__ lens,
                                                           const int32_t* __restrict__ cu, int64_t total_tokens,
                                                           int n_seq, int S_max, int NH, float scale,
                                                           float* __restrict__ out, uint32_t* fault) {
    constexpr int KB = 16;                       // keys per block
    constexpr int DG = D / 4;                    // dimensions per lane of a query row
    constexpr int NIT = KB * (D / 4) / 64;       // 16-byte pieces of a K (or V) block per lane
    static_assert(DG % 4 == 0 && NIT >= 1, "head dimension must be a multiple of 16");
    __shared__ __attribute__((aligned(16))) float Ks[KB][D];
    __shared__ __attribute__((aligned(16))) float Vs[KB][D];
    const int lane = threadIdx.x;
    const int qi = lane & 15, g = lane >> 4;
    const int jb = blockIdx.x >> 3;
    const int head = jb % NH;
    const int seq = (jb / NH) * 8 + (blockIdx.x & 7);
    if (seq >= n_seq) return;
    const int H = NH * D;
    const size_t tok = (size_t)3 * H;
    int64_t first;
    int S, len;
    if (cu != nullptr) {
        first = cu[seq];
        const int64_t n = (int64_t)cu[seq + 1] - first;
        if (first < 0 || n < 0 || n > S_max || first + n > total_tokens) {
            if (lane == 0) rag_flag(fault, RAG_FAULT_SEGMENT);
            return;
        }
        len = (int)n;
        S = len;
    } else {
        first = (int64_t)seq * S_max;
        S = S_max;
        len = min(max(lens[seq], 0), S);
    }
    const float* base = qkv + (size_t)first * tok + head * D;
    float* obase = out + (size_t)first * H + head * D;

    for (int q0 = 0; q0 < S; q0 += 16) {
        const int q = q0 + qi;
        const bool live = q < S && len > 0;
        float qv[DG], o[DG];
#pragma unroll
        for (int d = 0; d < DG; ++d) { qv[d] = 0.f; o[d] = 0.f; }
        if (live) {
#pragma unroll
            for (int d = 0; d < DG; d += 4) {
                const float4 v = *reinterpret_cast<const float4*>(base + (size_t)q * tok + g * DG + d);
                qv[d] = v.x * scale; qv[d + 1] = v.y * scale; qv[d + 2] = v.z * scale; qv[d + 3] = v.w * scale;
            }
        }
        float m_run = NEG_BIG, l_run = 0.f;
        for (int kb = 0; kb < len; kb += KB) {
            const int nk = min(KB, len - kb);
            float4 kreg[NIT], vreg[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = lane + 64 * it;
                const int key = i / (D / 4), c = i - key * (D / 4);
                kreg[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                vreg[it] = kreg[it];
                if (key < nk) {                           // rows beyond the sequence read as zero
                    const float* p = base + (size_t)(kb + key) * tok + c * 4;
                    kreg[it] = *reinterpret_cast<const float4*>(p + H);
                    vreg[it] = *reinterpret_cast<const float4*>(p + 2 * H);
                }
            }
            __syncthreads();                              // the previous block has been consumed
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = lane + 64 * it;
                const int key = i / (D / 4), c = i - key * (D / 4);
                *reinterpret_cast<float4*>(&Ks[key][c * 4]) = kreg[it];
                *reinterpret_cast<float4*>(&Vs[key][c * 4]) = vreg[it];
            }
            __syncthreads();
            float sc[KB];
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                float a = 0.f;
#pragma unroll
                for (int d = 0; d < DG; d += 4) {
                    const float4 kv = *reinterpret_cast<const float4*>(&Ks[k][g * DG + d]);
                    a = fmaf(qv[d], kv.x, a); a = fmaf(qv[d + 1], kv.y, a);
                    a = fmaf(qv[d + 2], kv.z, a); a = fmaf(qv[d + 3], kv.w, a);
                }
                sc[k] = a;
            }
            float mx = NEG_BIG;
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                float a = sc[k];
                a += __shfl_xor(a, 16, 64);
                a += __shfl_xor(a, 32, 64);               // every lane of the row now holds the same sum
                sc[k] = k < nk ? a : NEG_BIG;
                mx = fmaxf(mx, sc[k]);
            }
            const float m_new = fmaxf(m_run, mx);
            const float alpha = expf(m_run - m_new);
#pragma unroll
            for (int d = 0; d < DG; ++d) o[d] *= alpha;
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const float pk = k < nk ? expf(sc[k] - m_new) : 0.f;
                sum += pk;
#pragma unroll
                for (int d = 0; d < DG; d += 4) {
                    const float4 vv = *reinterpret_cast<const float4*>(&Vs[k][g * DG + d]);
                    o[d] = fmaf(pk, vv.x, o[d]); o[d + 1] = fmaf(pk, vv.y, o[d + 1]);
                    o[d + 2] = fmaf(pk, vv.z, o[d + 2]); o[d + 3] = fmaf(pk, vv.w, o[d + 3]);
                }
            }
            l_run = l_run * alpha + sum;
            m_run = m_new;
        }
        if (q < S) {
            const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
#pragma unroll
            for (int d = 0; d < DG; d += 4)
                *reinterpret_cast<float4*>(obase + (size_t)q * H + g * DG + d) =
                    make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
        }
    }
}

// Mean over each sequence's tokens, then (optionally) L2 normalisation: one wave per sequence.
__global__ __launch_bounds__(256) void mean_pool_kernel(const float* __restrict__ x, const int32_t* __restrict__ cu,
                                                        int64_t total_tokens, int n_seq, int H, int normalize,
                                                        float* __restrict__ out, uint32_t* fault) {
    const int lane = threadIdx.x & 63;
    const int seq = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seq >= n_seq) return;
    const int64_t a = cu[seq], b = cu[seq + 1];
    const int n_chunks = H / 8;
    float v[LN_MAX_CHUNKS][8];
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
    if (a < 0 || b < a || b > total_tokens) {
        if (lane == 0) rag_flag(fault, RAG_FAULT_SEGMENT);
    } else {
        for (int64_t t = a; t < b; ++t) {
#pragma unroll
            for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
                const int ch = lane + 64 * c;
                if (ch < n_chunks) {
                    float f[8];
                    Vec8<float>::load(x + t * H + ch * 8, f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[c][j] += f[j];
                }
            }
        }
    }
    const float inv_n = b > a ? 1.f / (float)(b - a) : 0.f;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[c][j] *= inv_n; sq += v[c][j] * v[c][j]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    const float sc = normalize ? 1.f / fmaxf(sqrtf(sq), 1e-12f) : 1.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int ch = lane + 64 * c;
        if (ch < n_chunks) {
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = v[c][j] * sc;
            Vec8<float>::store(out + (size_t)seq * H + ch * 8, y);
        }
    }
}

// ---- LayerNorm(x + residual) ---------------------------------------------------------------

template <typename T>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                            const T* __restrict__ gamma, const T* __restrict__ beta,
                                                            float eps, int64_t rows, int H, T* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int n_chunks = H / 8;
    float v[LN_MAX_CHUNKS][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int ch = lane + 64 * c;
        if (ch < n_chunks) {
            Vec8<T>::load(x + row * H + ch * 8, v[c]);
            if (res != nullptr) {
                float rr[8];
                Vec8<T>::load(res + row * H + ch * 8, rr);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[c][j] += rr[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[c][j];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float mean = sum / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int ch = lane + 64 * c;
        if (ch < n_chunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[c][j] - mean; sq += d * d; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    const float rstd = rsqrtf(sq / (float)H + eps);
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int ch = lane + 64 * c;
        if (ch < n_chunks) {
            float g[8], b[8], y[8];
            Vec8<T>::load(gamma + ch * 8, g);
            Vec8<T>::load(beta + ch * 8, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = (v[c][j] - mean) * rstd * g[j] + b[j];
            Vec8<T>::store(out + row * H + ch * 8, y);
        }
    }
}

// LayerNorm((y + bias) + residual) in fp32, written as fp32 and, for the split-precision GEMM that reads it
// next, as the bf16 row [hi | hi | lo] as well (out_split may be null).  One wave per row.
template <typename H16>
__global__ __launch_bounds__(256) void bias_add_layernorm_split_kernel(const float* __restrict__ y,
                                                                       const float* __restrict__ bias,
                                                                       const float* __restrict__ res,
                                                                       const float* __restrict__ gamma,
                                                                       const float* __restrict__ beta, float eps,
                                                                       int64_t rows, int H, float* __restrict__ out,
                                                                       H16* __restrict__ out_split) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int n_chunks = H / 8;
    float v[LN_MAX_CHUNKS][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int ch = lane + 64 * c;
        if (ch < n_chunks) {
            float bb[8], rr[8];
            Vec8<float>::load(y + row * H + ch * 8, v[c]);
            Vec8<float>::load(bias + ch * 8, bb);
            Vec8<float>::load(res + row * H + ch * 8, rr);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[c][j] = (v[c][j] + bb[j]) + rr[j];
                sum += v[c][j];
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float mean = sum / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int ch = lane + 64 * c;
        if (ch < n_chunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[c][j] - mean; sq += d * d; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    const float rstd = rsqrtf(sq / (float)H + eps);
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int ch = lane + 64 * c;
        if (ch < n_chunks) {
            float g[8], b[8], o[8];
            Vec8<float>::load(gamma + ch * 8, g);
            Vec8<float>::load(beta + ch * 8, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[c][j] - mean) * rstd * g[j] + b[j];
            Vec8<float>::store(out + row * H + ch * 8, o);
            if (out_split != nullptr) {
                typename Half16<H16>::x8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    RAG_SPLIT_HI_LO(H16, o[j], hi[j], lo[j]);
                }
                H16* p = out_split + row * 3 * H + ch * 8;
                *reinterpret_cast<typename Half16<H16>::x8*>(p) = hi;
                *reinterpret_cast<typename Half16<H16>::x8*>(p + H) = hi;
                *reinterpret_cast<typename Half16<H16>::x8*>(p + 2 * H) = lo;
            }
        }
    }
}

// LayerNorm(word[id] + position[pos] + type[tt]) per token, one wave per token.
// SPLIT (float tables only): also write the row as [hi | hi | lo] in that 16-bit type, the left operand of the
// first split-precision GEMM (what split_bf16x3_kernel would make of the row in a pass of its own).
template <typename T, typename SPLIT = bf16>
__global__ __launch_bounds__(256) void embed_layernorm_kernel(const int32_t* __restrict__ ids,
                                                              const int32_t* __restrict__ types,
                                                              const int32_t* __restrict__ pos, const T* __restrict__ word,
                                                              const T* __restrict__ position, const T* __restrict__ type_emb,
                                                              const T* __restrict__ gamma, const T* __restrict__ beta,
                                                              float eps, int64_t n_tokens, int H, int vocab, int max_pos,
                                                              int n_types, T* __restrict__ out, uint32_t* fault,
                                                              SPLIT* __restrict__ split = nullptr) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_tokens) return;
    int id = ids[t], ps = pos[t], tt = types != nullptr ? types[t] : 0;
    if (id < 0 || id >= vocab || ps < 0 || ps >= max_pos || tt < 0 || tt >= n_types) {
        if (lane == 0) rag_flag(fault, RAG_FAULT_TOKEN);     // outside the tables: embed as entry 0
        id = min(max(id, 0), vocab - 1);
        ps = min(max(ps, 0), max_pos - 1);
        tt = min(max(tt, 0), n_types - 1);
    }
    const int n_chunks = H / 8;
    float v[LN_MAX_CHUNKS][8];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int ch = lane + 64 * c;
        if (ch < n_chunks) {
            float a[8], b[8];
            Vec8<T>::load(word + (size_t)id * H + ch * 8, v[c]);
            Vec8<T>::load(position + (size_t)ps * H + ch * 8, a);
            Vec8<T>::load(type_emb + (size_t)tt * H + ch * 8, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[c][j] += a[j] + b[j]; sum += v[c][j]; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const float mean = sum / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int ch = lane + 64 * c;
        if (ch < n_chunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[c][j] - mean; sq += d * d; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    const float rstd = rsqrtf(sq / (float)H + eps);
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
        const int ch = lane + 64 * c;
        if (ch < n_chunks) {
            float g[8], b[8], y[8];
            Vec8<T>::load(gamma + ch * 8, g);
            Vec8<T>::load(beta + ch * 8, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = (v[c][j] - mean) * rstd * g[j] + b[j];
            if (out != nullptr) Vec8<T>::store(out + t * H + ch * 8, y);
            if constexpr (std::is_same<T, float>::value) {
                if (split != nullptr) {
                    typename Half16<SPLIT>::x8 hi, lo;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        RAG_SPLIT_HI_LO(SPLIT, y[j], hi[j], lo[j]);
                    }
                    SPLIT* o = split + t * 3 * H + ch * 8;
                    *reinterpret_cast<typename Half16<SPLIT>::x8*>(o) = hi;
                    *reinterpret_cast<typename Half16<SPLIT>::x8*>(o + H) = hi;
                    *reinterpret_cast<typename Half16<SPLIT>::x8*>(o + 2 * H) = lo;
                }
            }
        }
    }
}

}  // namespace

namespace {
int attention_launch(const void* d_qkv, const int32_t* d_lens, const int32_t* d_cu, int64_t total_tokens, int32_t n_seq,
                     int32_t seq_len, int32_t n_heads, int32_t head_dim, int32_t dtype, void* d_out, void* stream_,
                     int first_only = 0);
}

extern "C" int rag_enc_attention_dev(const void* d_qkv, const int32_t* d_lens, int32_t n_seq, int32_t seq_len,
                                     int32_t n_heads, int32_t head_dim, int32_t dtype, void* d_out, void* stream_) {
    RAG_REQUIRE(d_lens != nullptr || n_seq == 0, RAG_EINVAL, "rag_enc_attention: null lengths");
    return attention_launch(d_qkv, d_lens, nullptr, (int64_t)n_seq * seq_len, n_seq, seq_len, n_heads, head_dim, dtype,
                            d_out, stream_);
}

extern "C" int rag_enc_attention_packed_dev(const void* d_qkv, const int32_t* d_cu_seqlens, int32_t n_seq,
                                            int64_t total_tokens, int32_t max_seq_len, int32_t n_heads,
                                            int32_t head_dim, int32_t dtype, void* d_out, void* stream_) {
    RAG_REQUIRE(d_cu_seqlens != nullptr || n_seq == 0, RAG_EINVAL, "rag_enc_attention_packed: null offsets");
    RAG_REQUIRE(total_tokens >= 0 && total_tokens < (1ll << 31), RAG_ELIMIT, "rag_enc_attention_packed: bad token count");
    return attention_launch(d_qkv, nullptr, d_cu_seqlens, total_tokens, n_seq, max_seq_len, n_heads, head_dim, dtype,
                            d_out, stream_);
}

extern "C" int rag_enc_attention_packed_first_dev(const void* d_qkv, const int32_t* d_cu_seqlens, int32_t n_seq,
                                                  int64_t total_tokens, int32_t max_seq_len, int32_t n_heads,
                                                  int32_t head_dim, int32_t dtype, void* d_out, void* stream_) {
    RAG_REQUIRE(d_cu_seqlens != nullptr || n_seq == 0, RAG_EINVAL, "rag_enc_attention_packed_first: null offsets");
    RAG_REQUIRE(total_tokens >= 0 && total_tokens < (1ll << 31), RAG_ELIMIT,
                "rag_enc_attention_packed_first: bad token count");
    return attention_launch(d_qkv, nullptr, d_cu_seqlens, total_tokens, n_seq, max_seq_len, n_heads, head_dim, dtype,
                            d_out, stream_, 1);
}

namespace {
int attention_split_launch(const float* d_qkv, const float* d_bias, const int32_t* d_cu_seqlens, int32_t n_seq,
                           int64_t total_tokens, int32_t max_seq_len, int32_t n_heads, int32_t head_dim,
                           int32_t first_only, int32_t half_dtype, int32_t out_hl, void* d_out, void* stream_) {
    RAG_REQUIRE(half_dtype == RAG_DTYPE_BF16 || half_dtype == RAG_DTYPE_F16, RAG_EINVAL,
                "rag_enc_attention_split: the halves are bf16 or fp16");
    RAG_REQUIRE(n_seq >= 0 && max_seq_len >= 1 && n_heads >= 1, RAG_EINVAL, "rag_enc_attention_split: bad sizes");
    RAG_REQUIRE(head_dim == 32 || head_dim == 64, RAG_ELIMIT, "rag_enc_attention_split: head_dim must be 32 or 64 (got %d)",
                head_dim);
    RAG_REQUIRE(total_tokens >= 0 && total_tokens < (1ll << 31), RAG_ELIMIT, "rag_enc_attention_split: bad token count");
    if (n_seq == 0) return RAG_OK;
    RAG_REQUIRE(d_qkv && d_cu_seqlens && d_out, RAG_EINVAL, "rag_enc_attention_split: null pointer");
    RAG_REQUIRE(reinterpret_cast<uintptr_t>(d_qkv) % 16 == 0 && reinterpret_cast<uintptr_t>(d_bias) % 16 == 0 &&
                reinterpret_cast<uintptr_t>(d_out) % 16 == 0, RAG_EINVAL, "rag_enc_attention_split: pointers must be 16-byte aligned");
    RAG_REQUIRE(((int64_t)n_seq + 8) * n_heads < (1ll << 31), RAG_ELIMIT, "rag_enc_attention_split: too many (sequence, head) pairs");
    const int kpad = (max_seq_len + 31) & ~31;
    const size_t lds = head_dim == 32 ? (size_t)kpad * 256         // four images of 64-byte rows
                                      : 2 * ((size_t)kpad * (head_dim + 8) * 2 + (size_t)head_dim * (kpad + 4) * 2);
    RAG_REQUIRE(lds <= 160 * 1024, RAG_ELIMIT, "rag_enc_attention_split: sequences of %d tokens with head size %d need %zu bytes of LDS",
                max_seq_len, head_dim, lds);
    const float scale = 1.4426950408889634f / sqrtf((float)head_dim);
    const unsigned grid = (unsigned)((int64_t)((n_seq + 7) / 8) * 8 * n_heads);
    const int nt = first_only ? ATT_NT : 64 * std::min(4, (max_seq_len + 31) / 32);
    hipStream_t stream = (hipStream_t)stream_;
#define RAG_ATT_SPLIT_LAUNCH(DD, HH)                                                                                  \
    do {                                                                                                               \
        if (lds > 64 * 1024) RAG_GRANT_LDS((attention_split_kernel<DD, HH>), 160 * 1024);                              \
        attention_split_kernel<DD, HH><<<grid, nt, lds, stream>>>(d_qkv, d_bias, d_cu_seqlens, total_tokens, n_seq,    \
                                                                  max_seq_len, n_heads, scale, first_only, out_hl,     \
                                                                  (HH*)d_out, rag_fault_word());                       \
    } while (0)
    if (head_dim == 32 && half_dtype == RAG_DTYPE_BF16) RAG_ATT_SPLIT_LAUNCH(32, bf16);
    else if (head_dim == 32) RAG_ATT_SPLIT_LAUNCH(32, f16);
    else if (half_dtype == RAG_DTYPE_BF16) RAG_ATT_SPLIT_LAUNCH(64, bf16);
    else RAG_ATT_SPLIT_LAUNCH(64, f16);
#undef RAG_ATT_SPLIT_LAUNCH
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}
}  // namespace

extern "C" int rag_enc_attention_split_dev(const float* d_qkv, const float* d_bias, const int32_t* d_cu_seqlens,
                                           int32_t n_seq, int64_t total_tokens, int32_t max_seq_len, int32_t n_heads,
                                           int32_t head_dim, int32_t first_only, int32_t half_dtype, void* d_out,
                                           void* stream_) {
    return attention_split_launch(d_qkv, d_bias, d_cu_seqlens, n_seq, total_tokens, max_seq_len, n_heads, head_dim,
                                  first_only, half_dtype, 0, d_out, stream_);
}

extern "C" int rag_enc_attention_split_hl_dev(const float* d_qkv, const float* d_bias, const int32_t* d_cu_seqlens,
                                              int32_t n_seq, int64_t total_tokens, int32_t max_seq_len, int32_t n_heads,
                                              int32_t head_dim, int32_t first_only, int32_t half_dtype, void* d_out,
                                              void* stream_) {
    return attention_split_launch(d_qkv, d_bias, d_cu_seqlens, n_seq, total_tokens, max_seq_len, n_heads, head_dim,
                                  first_only, half_dtype, 1, d_out, stream_);
}

namespace {
int attention_launch(const void* d_qkv, const int32_t* d_lens, const int32_t* d_cu, int64_t total_tokens, int32_t n_seq,
                     int32_t seq_len, int32_t n_heads, int32_t head_dim, int32_t dtype, void* d_out, void* stream_,
                     int first_only) {
    RAG_REQUIRE(!first_only || dtype != RAG_DTYPE_F32, RAG_ELIMIT, "rag_enc_attention: first-token form is bf16 / fp16 only");
    RAG_REQUIRE(n_seq >= 0 && seq_len >= 1 && n_heads >= 1, RAG_EINVAL, "rag_enc_attention: bad sizes");
    RAG_REQUIRE(dtype == RAG_DTYPE_BF16 || dtype == RAG_DTYPE_F16 || dtype == RAG_DTYPE_F32, RAG_EINVAL,
                "rag_enc_attention: unknown dtype");
    RAG_REQUIRE(head_dim == 32 || head_dim == 64, RAG_ELIMIT, "rag_enc_attention: head_dim must be 32 or 64 (got %d)",
                head_dim);
    RAG_REQUIRE(seq_len <= RAG_ENC_MAX_SEQ, RAG_ELIMIT, "rag_enc_attention: seq_len %d exceeds %d", seq_len,
                RAG_ENC_MAX_SEQ);
    if (n_seq == 0) return RAG_OK;
    RAG_REQUIRE(d_qkv && d_out, RAG_EINVAL, "rag_enc_attention: null pointer");
    RAG_REQUIRE(((int64_t)n_seq + 8) * n_heads < (1ll << 31), RAG_ELIMIT, "rag_enc_attention: too many (sequence, head) pairs");
    hipStream_t stream = (hipStream_t)stream_;
    const int kpad = (seq_len + 31) & ~31;
    const size_t lds = (size_t)kpad * (head_dim + 8) * 2 + (size_t)head_dim * (kpad + 4) * 2;
    const float scale = 1.4426950408889634f / sqrtf((float)head_dim);   // log2(e) / sqrt(D): the kernel works in base 2
    const unsigned grid = (unsigned)((int64_t)((n_seq + 7) / 8) * 8 * n_heads);
    // a wave per 32-query block of the longest sequence, four at most: waves without a block of their own
    // would only hold wave slots that other workgroups of the compute unit can use
    const int att_nt = first_only ? ATT_NT : 64 * std::min(4, (seq_len + 31) / 32);
    if (dtype == RAG_DTYPE_F32) {
        const float scale32 = 1.0f / sqrtf((float)head_dim);
        if (head_dim == 32)
            attention_f32_kernel<32><<<grid, 64, 0, (hipStream_t)stream_>>>((const float*)d_qkv, d_lens, d_cu, total_tokens,
                                                                           n_seq, seq_len, n_heads, scale32,
                                                                           (float*)d_out, rag_fault_word());
        else
            attention_f32_kernel<64><<<grid, 64, 0, (hipStream_t)stream_>>>((const float*)d_qkv, d_lens, d_cu, total_tokens,
                                                                           n_seq, seq_len, n_heads, scale32,
                                                                           (float*)d_out, rag_fault_word());
        RAG_HIP_TRY(hipGetLastError());
        return RAG_OK;
    }
#define RAG_ATT_LAUNCH(D_, E_)                                                                                          \
    do {                                                                                                                \
        if (lds > 64 * 1024) RAG_GRANT_LDS((attention_kernel<D_, E_>), 160 * 1024);                                     \
        attention_kernel<D_, E_><<<grid, att_nt, lds, stream>>>((const E_*)d_qkv, d_lens, d_cu, total_tokens, n_seq,    \
                                                                seq_len, n_heads, scale, first_only, (E_*)d_out,        \
                                                                rag_fault_word());                                      \
    } while (0)
    if (head_dim == 32 && dtype == RAG_DTYPE_BF16) RAG_ATT_LAUNCH(32, bf16);
    else if (head_dim == 32) RAG_ATT_LAUNCH(32, f16);
    else if (dtype == RAG_DTYPE_BF16) RAG_ATT_LAUNCH(64, bf16);
    else RAG_ATT_LAUNCH(64, f16);
#undef RAG_ATT_LAUNCH
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}
}  // namespace

extern "C" int rag_enc_add_layernorm_dev(const void* d_x, const void* d_residual, const void* d_gamma,
                                         const void* d_beta, float eps, int64_t rows, int32_t hidden, int32_t dtype,
                                         void* d_out, void* stream_) {
    RAG_REQUIRE(rows >= 0 && hidden >= 8 && hidden % 8 == 0 && hidden <= 8 * 64 * LN_MAX_CHUNKS, RAG_ELIMIT,
                "rag_enc_add_layernorm: hidden must be a multiple of 8 up to %d (got %d)", 8 * 64 * LN_MAX_CHUNKS, hidden);
    RAG_REQUIRE(dtype == RAG_DTYPE_BF16 || dtype == RAG_DTYPE_F16 || dtype == RAG_DTYPE_F32, RAG_EINVAL,
                "rag_enc_add_layernorm: unknown dtype");
    if (rows == 0) return RAG_OK;
    RAG_REQUIRE(d_x && d_gamma && d_beta && d_out, RAG_EINVAL, "rag_enc_add_layernorm: null pointer");
    RAG_REQUIRE((rows + 3) / 4 < (1ll << 31), RAG_ELIMIT, "rag_enc_add_layernorm: too many rows");
    hipStream_t stream = (hipStream_t)stream_;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    if (dtype == RAG_DTYPE_BF16)
        add_layernorm_kernel<bf16><<<grid, 256, 0, stream>>>((const bf16*)d_x, (const bf16*)d_residual,
                                                             (const bf16*)d_gamma, (const bf16*)d_beta, eps, rows, hidden,
                                                             (bf16*)d_out);
    else if (dtype == RAG_DTYPE_F16)
        add_layernorm_kernel<f16><<<grid, 256, 0, stream>>>((const f16*)d_x, (const f16*)d_residual, (const f16*)d_gamma,
                                                            (const f16*)d_beta, eps, rows, hidden, (f16*)d_out);
    else
        add_layernorm_kernel<float><<<grid, 256, 0, stream>>>((const float*)d_x, (const float*)d_residual,
                                                              (const float*)d_gamma, (const float*)d_beta, eps, rows,
                                                              hidden, (float*)d_out);
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

namespace {
int embed_layernorm_launch(const int32_t* d_ids, const int32_t* d_types, const int32_t* d_pos, const void* d_word,
                           const void* d_position, const void* d_type_emb, const void* d_gamma, const void* d_beta, float eps,
                           int64_t n_tokens, int32_t hidden, int32_t vocab, int32_t max_pos, int32_t n_types, int32_t dtype,
                           void* d_out, int32_t half_dtype, void* d_out_split, void* stream_) {
    RAG_REQUIRE(n_tokens >= 0 && hidden >= 8 && hidden % 8 == 0 && hidden <= 8 * 64 * LN_MAX_CHUNKS, RAG_ELIMIT,
                "rag_enc_embed_layernorm: hidden must be a multiple of 8 up to %d (got %d)", 8 * 64 * LN_MAX_CHUNKS, hidden);
    RAG_REQUIRE(vocab >= 1 && max_pos >= 1 && n_types >= 1, RAG_EINVAL, "rag_enc_embed_layernorm: empty table");
    RAG_REQUIRE(dtype == RAG_DTYPE_BF16 || dtype == RAG_DTYPE_F16 || dtype == RAG_DTYPE_F32, RAG_EINVAL,
                "rag_enc_embed_layernorm: unknown dtype");
    RAG_REQUIRE(d_out_split == nullptr || (dtype == RAG_DTYPE_F32 && (half_dtype == RAG_DTYPE_BF16 || half_dtype == RAG_DTYPE_F16)),
                RAG_EINVAL, "rag_enc_embed_layernorm_split: fp32 tables and bf16 or fp16 halves");
    if (n_tokens == 0) return RAG_OK;
    RAG_REQUIRE(d_ids && d_pos && d_word && d_position && d_type_emb && d_gamma && d_beta && (d_out || d_out_split), RAG_EINVAL,
                "rag_enc_embed_layernorm: null pointer");
    RAG_REQUIRE((n_tokens + 3) / 4 < (1ll << 31), RAG_ELIMIT, "rag_enc_embed_layernorm: too many tokens");
    hipStream_t stream = (hipStream_t)stream_;
    const unsigned grid = (unsigned)((n_tokens + 3) / 4);
    if (dtype == RAG_DTYPE_BF16)
        embed_layernorm_kernel<bf16><<<grid, 256, 0, stream>>>(d_ids, d_types, d_pos, (const bf16*)d_word,
                                                               (const bf16*)d_position, (const bf16*)d_type_emb,
                                                               (const bf16*)d_gamma, (const bf16*)d_beta, eps, n_tokens,
                                                               hidden, vocab, max_pos, n_types, (bf16*)d_out,
                                                               rag_fault_word());
    else if (dtype == RAG_DTYPE_F16)
        embed_layernorm_kernel<f16><<<grid, 256, 0, stream>>>(d_ids, d_types, d_pos, (const f16*)d_word,
                                                              (const f16*)d_position, (const f16*)d_type_emb,
                                                              (const f16*)d_gamma, (const f16*)d_beta, eps, n_tokens,
                                                              hidden, vocab, max_pos, n_types, (f16*)d_out,
                                                              rag_fault_word());
    else if (d_out_split != nullptr && half_dtype == RAG_DTYPE_F16)
        embed_layernorm_kernel<float, _Float16><<<grid, 256, 0, stream>>>(
            d_ids, d_types, d_pos, (const float*)d_word, (const float*)d_position, (const float*)d_type_emb,
            (const float*)d_gamma, (const float*)d_beta, eps, n_tokens, hidden, vocab, max_pos, n_types, (float*)d_out,
            rag_fault_word(), (_Float16*)d_out_split);
    else
        embed_layernorm_kernel<float, bf16><<<grid, 256, 0, stream>>>(
            d_ids, d_types, d_pos, (const float*)d_word, (const float*)d_position, (const float*)d_type_emb,
            (const float*)d_gamma, (const float*)d_beta, eps, n_tokens, hidden, vocab, max_pos, n_types, (float*)d_out,
            rag_fault_word(), (bf16*)d_out_split);
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}
}  // namespace

extern "C" int rag_enc_embed_layernorm_dev(const int32_t* d_ids, const int32_t* d_types, const int32_t* d_pos,
                                           const void* d_word, const void* d_position, const void* d_type_emb,
                                           const void* d_gamma, const void* d_beta, float eps, int64_t n_tokens,
                                           int32_t hidden, int32_t vocab, int32_t max_pos, int32_t n_types,
                                           int32_t dtype, void* d_out, void* stream_) {
    return embed_layernorm_launch(d_ids, d_types, d_pos, d_word, d_position, d_type_emb, d_gamma, d_beta, eps, n_tokens,
                                  hidden, vocab, max_pos, n_types, dtype, d_out, 0, nullptr, stream_);
}

extern "C" int rag_enc_embed_layernorm_split_dev(const int32_t* d_ids, const int32_t* d_types, const int32_t* d_pos,
                                                 const float* d_word, const float* d_position, const float* d_type_emb,
                                                 const float* d_gamma, const float* d_beta, float eps, int64_t n_tokens,
                                                 int32_t hidden, int32_t vocab, int32_t max_pos, int32_t n_types,
                                                 float* d_out, int32_t half_dtype, void* d_out_split, void* stream_) {
    RAG_REQUIRE(d_out_split != nullptr || n_tokens == 0, RAG_EINVAL, "rag_enc_embed_layernorm_split: null split output");
    return embed_layernorm_launch(d_ids, d_types, d_pos, d_word, d_position, d_type_emb, d_gamma, d_beta, eps, n_tokens,
                                  hidden, vocab, max_pos, n_types, RAG_DTYPE_F32, d_out, half_dtype, d_out_split, stream_);
}

namespace {
// x (fp32) = hi + lo + (what is left below 2^-17 relative), hi and lo in bf16; one row of the
// output is [hi | hi | lo], the left operand of a GEMM against [w_hi ; w_lo ; w_hi].
template <typename H16>
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ x, int64_t rows, int K,
                                                           H16* __restrict__ out) {
    const int64_t n8 = rows * (K / 8);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / (K / 8);
        const int c = (int)(i - row * (K / 8));
        float f[8];
        Vec8<float>::load(x + row * K + c * 8, f);
        typename Half16<H16>::x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            RAG_SPLIT_HI_LO(H16, f[j], hi[j], lo[j]);
        }
        H16* o = out + row * 3 * K + c * 8;
        *reinterpret_cast<typename Half16<H16>::x8*>(o) = hi;
        *reinterpret_cast<typename Half16<H16>::x8*>(o + K) = hi;
        *reinterpret_cast<typename Half16<H16>::x8*>(o + 2 * K) = lo;
    }
}
}  // namespace

namespace {
// The same split applied to act(y + bias): what stands between two split-precision GEMMs of the feed-forward
// block (bias add, GELU and split were three passes over the widest activation of the layer).
template <bool GELU, typename H16>
__global__ __launch_bounds__(256) void bias_act_split_bf16x3_kernel(const float* __restrict__ y,
                                                                    const float* __restrict__ bias, int64_t rows, int K,
                                                                    H16* __restrict__ out) {
    const int64_t n8 = rows * (K / 8);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / (K / 8);
        const int c = (int)(i - row * (K / 8));
        float f[8], b[8];
        Vec8<float>::load(y + row * K + c * 8, f);
        Vec8<float>::load(bias + c * 8, b);
        typename Half16<H16>::x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f[j] += b[j];
            if (GELU) f[j] = 0.5f * f[j] * (1.0f + erff(f[j] * 0.70710678118654752f));
            RAG_SPLIT_HI_LO(H16, f[j], hi[j], lo[j]);
        }
        H16* o = out + row * 3 * K + c * 8;
        *reinterpret_cast<typename Half16<H16>::x8*>(o) = hi;
        *reinterpret_cast<typename Half16<H16>::x8*>(o + K) = hi;
        *reinterpret_cast<typename Half16<H16>::x8*>(o + 2 * K) = lo;
    }
}
}  // namespace

extern "C" int rag_enc_bias_act_split_bf16x3_dev(const float* d_y, const float* d_bias, int32_t gelu, int64_t rows,
                                                 int32_t k, int32_t half_dtype, void* d_out, void* stream_) {
    RAG_REQUIRE(half_dtype == RAG_DTYPE_BF16 || half_dtype == RAG_DTYPE_F16, RAG_EINVAL,
                "rag_enc_bias_act_split_bf16x3: the halves are bf16 or fp16");
    RAG_REQUIRE(rows >= 0 && k >= 8 && k % 8 == 0, RAG_EINVAL,
                "rag_enc_bias_act_split_bf16x3: k must be a positive multiple of 8");
    if (rows == 0) return RAG_OK;
    RAG_REQUIRE(d_y && d_bias && d_out, RAG_EINVAL, "rag_enc_bias_act_split_bf16x3: null pointer");
    const int64_t n8 = rows * (k / 8);
    const unsigned grid = (unsigned)(n8 / 256 + 1 > 65536 ? 65536 : n8 / 256 + 1);
    hipStream_t stream = (hipStream_t)stream_;
    if (half_dtype == RAG_DTYPE_BF16) {
        if (gelu) bias_act_split_bf16x3_kernel<true, bf16><<<grid, 256, 0, stream>>>(d_y, d_bias, rows, k, (bf16*)d_out);
        else bias_act_split_bf16x3_kernel<false, bf16><<<grid, 256, 0, stream>>>(d_y, d_bias, rows, k, (bf16*)d_out);
    } else {
        if (gelu) bias_act_split_bf16x3_kernel<true, f16><<<grid, 256, 0, stream>>>(d_y, d_bias, rows, k, (f16*)d_out);
        else bias_act_split_bf16x3_kernel<false, f16><<<grid, 256, 0, stream>>>(d_y, d_bias, rows, k, (f16*)d_out);
    }
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

extern "C" int rag_enc_split_bf16x3_dev(const float* d_x, int64_t rows, int32_t k, int32_t half_dtype, void* d_out,
                                        void* stream_) {
    RAG_REQUIRE(half_dtype == RAG_DTYPE_BF16 || half_dtype == RAG_DTYPE_F16, RAG_EINVAL,
                "rag_enc_split_bf16x3: the halves are bf16 or fp16");
    RAG_REQUIRE(rows >= 0 && k >= 8 && k % 8 == 0, RAG_EINVAL, "rag_enc_split_bf16x3: k must be a positive multiple of 8");
    if (rows == 0) return RAG_OK;
    RAG_REQUIRE(d_x && d_out, RAG_EINVAL, "rag_enc_split_bf16x3: null pointer");
    const int64_t n8 = rows * (k / 8);
    const unsigned grid = (unsigned)(n8 / 256 + 1 > 65536 ? 65536 : n8 / 256 + 1);
    if (half_dtype == RAG_DTYPE_BF16) split_bf16x3_kernel<bf16><<<grid, 256, 0, (hipStream_t)stream_>>>(d_x, rows, k, (bf16*)d_out);
    else split_bf16x3_kernel<f16><<<grid, 256, 0, (hipStream_t)stream_>>>(d_x, rows, k, (f16*)d_out);
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

extern "C" int rag_enc_bias_add_layernorm_split_dev(const float* d_y, const float* d_bias, const float* d_residual,
                                                   const float* d_gamma, const float* d_beta, float eps, int64_t rows,
                                                   int32_t hidden, int32_t half_dtype, float* d_out, void* d_out_split,
                                                   void* stream_) {
    RAG_REQUIRE(half_dtype == RAG_DTYPE_BF16 || half_dtype == RAG_DTYPE_F16, RAG_EINVAL,
                "rag_enc_bias_add_layernorm_split: the halves are bf16 or fp16");
    RAG_REQUIRE(rows >= 0 && hidden >= 8 && hidden % 8 == 0 && hidden <= 8 * 64 * LN_MAX_CHUNKS, RAG_ELIMIT,
                "rag_enc_bias_add_layernorm_split: hidden must be a multiple of 8 up to %d", 8 * 64 * LN_MAX_CHUNKS);
    if (rows == 0) return RAG_OK;
    RAG_REQUIRE(d_y && d_bias && d_residual && d_gamma && d_beta && d_out, RAG_EINVAL,
                "rag_enc_bias_add_layernorm_split: null pointer");
    RAG_REQUIRE((rows + 3) / 4 < (1ll << 31), RAG_ELIMIT, "rag_enc_bias_add_layernorm_split: too many rows");
    if (half_dtype == RAG_DTYPE_BF16)
        bias_add_layernorm_split_kernel<bf16><<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream_>>>(
            d_y, d_bias, d_residual, d_gamma, d_beta, eps, rows, hidden, d_out, (bf16*)d_out_split);
    else
        bias_add_layernorm_split_kernel<f16><<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream_>>>(
            d_y, d_bias, d_residual, d_gamma, d_beta, eps, rows, hidden, d_out, (f16*)d_out_split);
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

namespace {
// Exact (erf) GELU, the activation of BERT's feed-forward block, elementwise and in place.
template <typename T>
__global__ __launch_bounds__(256) void gelu_kernel(T* __restrict__ x, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float f[8];
        Vec8<T>::load(x + i * 8, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = 0.5f * f[j] * (1.0f + erff(f[j] * 0.70710678118654752f));
        Vec8<T>::store(x + i * 8, f);
    }
}
}  // namespace

extern "C" int rag_enc_gelu_dev(void* d_x, int64_t n, int32_t dtype, void* stream_) {
    RAG_REQUIRE(n >= 0 && n % 8 == 0, RAG_EINVAL, "rag_enc_gelu: element count must be a multiple of 8");
    RAG_REQUIRE(dtype == RAG_DTYPE_BF16 || dtype == RAG_DTYPE_F32, RAG_EINVAL, "rag_enc_gelu: unknown dtype");
    if (n == 0) return RAG_OK;
    RAG_REQUIRE(d_x != nullptr, RAG_EINVAL, "rag_enc_gelu: null pointer");
    const int64_t n8 = n / 8;
    const unsigned grid = (unsigned)(n8 / 256 + 1 > 256 * 64 ? 256 * 64 : n8 / 256 + 1);
    if (dtype == RAG_DTYPE_BF16) gelu_kernel<bf16><<<grid, 256, 0, (hipStream_t)stream_>>>((bf16*)d_x, n8);
    else gelu_kernel<float><<<grid, 256, 0, (hipStream_t)stream_>>>((float*)d_x, n8);
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

extern "C" int rag_enc_mean_pool_dev(const float* d_x, const int32_t* d_cu_seqlens, int32_t n_seq, int64_t total_tokens,
                                     int32_t hidden, int32_t normalize, float* d_out, void* stream_) {
    RAG_REQUIRE(n_seq >= 0 && total_tokens >= 0 && hidden >= 8 && hidden % 8 == 0 && hidden <= 8 * 64 * LN_MAX_CHUNKS,
                RAG_ELIMIT, "rag_enc_mean_pool: hidden must be a multiple of 8 up to %d", 8 * 64 * LN_MAX_CHUNKS);
    if (n_seq == 0) return RAG_OK;
    RAG_REQUIRE(d_x && d_cu_seqlens && d_out, RAG_EINVAL, "rag_enc_mean_pool: null pointer");
    mean_pool_kernel<<<(unsigned)((n_seq + 3) / 4), 256, 0, (hipStream_t)stream_>>>(d_x, d_cu_seqlens, total_tokens, n_seq,
                                                                                   hidden, normalize, d_out,
                                                                                   rag_fault_word());
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}
