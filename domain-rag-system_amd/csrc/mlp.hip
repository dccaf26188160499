// Feed-forward block of a 384-wide BERT layer (MiniLM-L6 cross-encoder) in ONE kernel:
//
//   Y = LayerNorm( X + GELU(X W1^T + b1) W2^T + b2 ) * gamma + beta
//
// X, Y [M][384] bf16 or fp16 (template parameter E); W1 [I][384], W2 [384][I] (nn.Linear layouts), fp32
// accumulation on v_mfma_f32_32x32x16_bf16 / _f16.  The [M][I] intermediate (I = 1536: four times the layer's input) never
// leaves the chip: per token the block reads 768 B and writes 768 B, where two GEMMs, a GELU pass
// and a residual+LayerNorm pass move 11.5 KB.  gfx950 only.
//
// One workgroup of four waves (one per SIMD, so each may use the whole 512-register file) owns
// 128 tokens; a wave owns 32 of them for the whole kernel.  Both products are computed
// transposed, the weights as the MFMA's A operand out of LDS and the activations as its B operand
// out of registers:
//   H^T[unit][token] = W1 . X^T      B fragments = the wave's 32 rows of X, loaded once (96 registers)
//   Y^T[feat][token] += W2 . H^T     B fragments = the accumulators of H^T, converted in place
// A lane of a 32x32 accumulator tile holds one token (its column) and rows (i&3) + 8(i>>2) + 4h:
// registers 8t..8t+7 are exactly the eight k values a lane must supply for k-step t of the next
// product if the rows of H^T are ordered to match, and that order is given to the W1 rows while
// they are staged (no shuffles, no LDS round trip for H).  The intermediate dimension is walked in
// chunks of 32 units: per chunk 24 MFMAs for H^T, the GELU on 16 registers, 24 MFMAs into the
// twelve 32x32 tiles of Y^T (192 registers), with three chunks in flight (first product of chunk
// c+2, GELU of c+1, second product of c).  W1 / W2 chunks are double buffered in LDS, filled by
// direct-to-LDS loads from L2 (the images are XOR-swizzled, not padded: conflict-free 16-byte
// reads); the fragment reads are issued from the source three k-steps ahead of their MFMAs.  The
// W2 rows are staged in the same swapped order, so Y^T comes out in the layout of the X fragments:
// the epilogue adds bias and the residual (the block's own input, still in registers), takes the
// LayerNorm statistics of a token from the lane and its partner in the other half-wave, and
// stores 16-bit values in 16-byte pieces.
//
// Behind CrossEncoderReranker.score_pairs (reference src/reranker/rerank.py:79-84): the
// intermediate / output sub-layers of the BertForSequenceClassification forward that
// CrossEncoder.predict runs.
#include "common.h"

#include <math.h>
#include <initializer_list>
#include <type_traits>
#include <utility>
#include <vector>

namespace {

#ifndef ML_STAMPS
#define ML_STAMPS 0
#endif
#if ML_STAMPS
__device__ unsigned long long ml_dbg[8];
#define ML_STAMP(var) unsigned long long var = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define ML_STAMP(var)
#endif

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16;
// the 16-bit element type of a kernel instance: bfloat16 (range of fp32, 8 significant bits) or float16 (11 bits,
// values below 65504), the same matrix-core rate either way
template <typename E> struct El16;
template <> struct El16<bf16> {
    typedef bf16 x8 __attribute__((ext_vector_type(8)));
    __device__ static __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct El16<f16> {
    typedef f16 x8 __attribute__((ext_vector_type(8)));
    __device__ static __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int ML_HID = 384;                  // layer width
constexpr int ML_BM = 128;                   // tokens per workgroup
constexpr int ML_NT = 256;                   // four waves
constexpr int ML_C = 32;                     // intermediate units per chunk
constexpr int ML_KS = ML_HID / 16;           // k-steps of the first product: 24
constexpr int ML_YT = ML_HID / 32;           // accumulator tiles of Y^T: 12
constexpr int ML_W_BYTES = ML_C * ML_HID * 2;             // one W1 chunk = one W2 chunk = 24576 B
constexpr int ML_W_DMAS = ML_W_BYTES / 1024 / (ML_NT / 64);   // 1-KiB direct-to-LDS loads per wave and chunk: 6
constexpr int ML_PF = 3;                    // fragment reads run this many k-steps ahead of their MFMAs
constexpr int ML_MAX_I = 2048;               // b1 sits in LDS (beside three W1 and three W2 buffers: 160 KB in all)
// LDS image: W1 buffers [3][32 rows][48 x 16 B], W2 buffers [3][384 rows][4 x 16 B] (rings of three: a chunk's weights
// are asked for two iterations before they are used), b1 [I] as fp32,
// then b2, gamma, beta [384] as bf16
#ifndef ML_DMA_EVERY
#define ML_DMA_EVERY 2      // one weight piece every this many k-steps (12 pieces per wave and chunk over 24 steps)
#endif
#ifndef ML_ABLATE_DMA
#define ML_ABLATE_DMA 0     // timing ablation only (wrong results): the loop without its weight loads
#endif
constexpr int ML_RING = 3;
constexpr int ML_OFF_W2 = ML_RING * ML_W_BYTES;
constexpr int ML_OFF_B1 = 2 * ML_RING * ML_W_BYTES;

template <typename E>
struct MlpParamsT {
    const E* x;        // [M][384]
    const E* w1;       // [I][384]
    const E* b1;       // [I]
    const E* w2;       // [384][I]
    const E* b2;       // [384]
    const E* gamma;    // [384]
    const E* beta;     // [384]
    E* y;              // [M][384]
    int64_t M;
    int I;
    float eps;
    // attention-output projection in front of the block (PROJ kernels): x above is then the layer's input
    const E* ctx;      // [M][384] attention output
    const E* wo;       // [384][384]
    const E* bo;       // [384]
    const E* gamma1;   // [384]
    const E* beta1;    // [384]
};

// One wave-instruction of a direct-to-LDS load: lane l's 16 bytes at gbase + voff land at LDS byte
// lds_dst + 16 l (the destination is wave-uniform base + lane order; only the source is per lane).
// M0 carries the destination and belongs to the compiler, so it is saved and restored inside the
// statement.  The compiler does not count this load: wait with s_waitcnt vmcnt before reading.
__device__ __forceinline__ void lds_dma16(uint32_t lds_dst, const void* gbase, uint32_t voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(gbase) : "memory");
}

// LDS reads whose completion the SOURCE waits for, so that they can run several MFMAs ahead of their use
// (the compiler issues a fragment read right before the MFMA that needs it and waits for it there).
// lds_read16 only issues; the value is defined after an lds_wait<N> that names the register, with
// N = the number of LDS reads issued after it (LDS reads complete in order).
template <int OFF, typename T>
__device__ __forceinline__ void lds_read16(T& dst, uint32_t addr) {
    static_assert(2 * ML_W_DMAS <= ML_KS, "one weight load per k-step");
    static_assert(sizeof(T) == 16 && OFF >= 0 && OFF < 65536, "16-byte read, 16-bit offset");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N, typename T>
__device__ __forceinline__ void lds_wait(T& a, T& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N, typename T>
__device__ __forceinline__ void lds_wait(T& a) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
}
template <int... Is, typename F>
__device__ __forceinline__ void unrolled(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}

// GELU(v) = v Phi(v), Phi(v) = (1 + erf(v / sqrt 2)) / 2, for two values at a time.  Phi(v) - 1/2 is an
// odd degree-17 polynomial in v fitted on |v| <= 4.525 and held constant beyond (Phi(4.525) = 1 - 3e-6):
// absolute error of the GELU below 6e-5, a fiftieth of the bf16 rounding of values near 1 (an eighth of the
// fp16 one) — and no
// exponential or reciprocal, which run at a quarter of the FMA rate.
// The same evaluation in three slices (head: clamp, square, three FMAs; mid: four FMAs; tail: the rest).
#define ML_K2(x) ((f32x2){x, x})
__device__ __forceinline__ void gelu_head(f32x2 v, f32x2& c, f32x2& s, f32x2& p) {
    c.x = __builtin_amdgcn_fmed3f(v.x, -4.525483400f, 4.525483400f);
    c.y = __builtin_amdgcn_fmed3f(v.y, -4.525483400f, 4.525483400f);
    s = c * c;
    p = __builtin_elementwise_fma(ML_K2(2.882145399e-11f), s, ML_K2(-3.184668485e-09f));
    p = __builtin_elementwise_fma(p, s, ML_K2(1.545971132e-07f));
    p = __builtin_elementwise_fma(p, s, ML_K2(-4.369227442e-06f));
}
__device__ __forceinline__ void gelu_mid(f32x2 s, f32x2& p) {
    p = __builtin_elementwise_fma(p, s, ML_K2(8.056942605e-05f));
    p = __builtin_elementwise_fma(p, s, ML_K2(-1.032623707e-03f));
    p = __builtin_elementwise_fma(p, s, ML_K2(9.570434947e-03f));
    p = __builtin_elementwise_fma(p, s, ML_K2(-6.596714181e-02f));
}
__device__ __forceinline__ f32x2 gelu_tail(f32x2 v, f32x2 c, f32x2 s, f32x2 p) {
    p = __builtin_elementwise_fma(p, s, ML_K2(3.987103009e-01f));
    const f32x2 phi = __builtin_elementwise_fma(p, c, ML_K2(0.5f));
    return v * phi;
}
__device__ __forceinline__ f32x2 gelu2(f32x2 v) {
    f32x2 c, s, p;
    gelu_head(v, c, s, p);
    gelu_mid(s, p);
    return gelu_tail(v, c, s, p);
}

// MODE 0: the feed-forward block of x.  MODE 1 (PROJ): x1 = LN1(x + ctx Wo^T + bo) first, then the block of x1.
// MODE 2: x1 = LN1(x + ctx) first, ctx being the finished attention-output projection (bias included): the
// residual + LayerNorm pass between the projection and this kernel, done on the rows this kernel loads anyway.
template <typename E, int MODE>
__global__ __launch_bounds__(ML_NT, 1) void mlp_fused_kernel(MlpParamsT<E> p) {
    typedef typename El16<E>::x8 e8;
    constexpr bool PROJ = MODE == 1;
    constexpr bool PRE_LN = MODE != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ML_STAMP(tk0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * ML_BM;
    const int64_t tok = m0 + wave * 32 + r;                   // this lane's token
    const int64_t tok_ld = min(tok, p.M - 1);                 // rows past M read the last row; never stored
    const int n_chunks = p.I / ML_C;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

    // ---- weight staging, direct to LDS.  A wave-instruction fills 1 KiB of the image in lane order, so
    // the image cannot be padded; bank conflicts are avoided by XOR-ing the 16-byte piece index with
    // row bits instead, applied here on the source address and again where the fragments are read.
    //   W1 image row q holds unit (q with bits 2 and 3 swapped) of the chunk: with that row order the
    //   H^T accumulator registers 8t..8t+7 of a lane are units 16t + 8h + 0..7, the natural k order of
    //   W2; piece j of the row sits at j ^ (q & 15).
    //   W2 image row f holds the chunk's 32 units of feature (f with bits 2 and 3 swapped), piece j at
    //   j ^ ((f >> 2) & 3).
    uint32_t off1[ML_W_DMAS], off2[ML_W_DMAS], offo[ML_W_DMAS];   // per-lane source byte offsets inside a chunk
#pragma unroll
    for (int i = 0; i < ML_W_DMAS; ++i) {
        const int L = 64 * (wave + 4 * i) + lane;             // piece index inside the image
        const int q = L / 48, j1 = (L % 48) ^ (q & 15);
        const int unit = (q & ~12) | ((q & 4) << 1) | ((q & 8) >> 1);
        off1[i] = (uint32_t)((unit * ML_HID + j1 * 8) * 2);
        // W2 and Wo (PROJ) images: row f holds output feature (f with bits 2 and 3 swapped), so that the
        // accumulators of Y^T — and of the projection — come out in the layout of the X fragments
        const int f = L / 4, j2 = (L % 4) ^ ((f >> 2) & 3);
        const int fo = (f & ~12) | ((f & 4) << 1) | ((f & 8) >> 1);
        off2[i] = (uint32_t)((fo * p.I + j2 * 8) * 2);
        offo[i] = (uint32_t)((fo * ML_HID + j2 * 8) * 2);
    }
    auto dma_w1 = [&](int c, int buf) {                      // chunk c of W1 -> W1 buffer buf
        const E* g = p.w1 + (size_t)c * ML_C * ML_HID;
#pragma unroll
        for (int i = 0; i < ML_W_DMAS; ++i) lds_dma16(lds0 + buf * ML_W_BYTES + 1024 * (wave + 4 * i), g, off1[i]);
    };
    auto dma_w2 = [&](int c, int buf) {
        const E* g = p.w2 + c * ML_C;
#pragma unroll
        for (int i = 0; i < ML_W_DMAS; ++i)
            lds_dma16(lds0 + ML_OFF_W2 + buf * ML_W_BYTES + 1024 * (wave + 4 * i), g, off2[i]);
    };
    const int last = n_chunks - 1;
    dma_w1(0, 0);
    dma_w1(min(1, last), 1);
    dma_w1(min(2, last), 2);
    if constexpr (!PROJ) {               // PROJ: the W2 buffers carry Wo first
        dma_w2(0, 0);
        dma_w2(min(1, last), 1);
    }
    for (int i = tid; i < p.I; i += ML_NT) *reinterpret_cast<float*>(smem + ML_OFF_B1 + 4 * i) = (float)p.b1[i];
    for (int i = tid; i < (PRE_LN ? 6 : 3) * (ML_HID / 8); i += ML_NT) {   // b2 | gamma | beta | bo | gamma1 | beta1 behind b1, 16-bit
        const int vsel = i / (ML_HID / 8), piece = i % (ML_HID / 8);
        const E* src = vsel == 0 ? p.b2 : vsel == 1 ? p.gamma : vsel == 2 ? p.beta : vsel == 3 ? (PROJ ? p.bo : p.b2) : vsel == 4 ? p.gamma1 : p.beta1;
        *reinterpret_cast<e8*>(smem + ML_OFF_B1 + 4 * (size_t)p.I + 2 * (vsel * ML_HID + piece * 8)) =
            *reinterpret_cast<const e8*>(src + piece * 8);
    }

    // ---- the wave's 32 rows of X as B fragments: lane (r, h) holds X[tok][16t + 8h .. + 7]
    // (PROJ: the rows of the attention output first; X is then computed below)
    e8 xf[ML_KS];
    {
        const E* xrow = (PRE_LN ? p.ctx : p.x) + tok_ld * ML_HID + 8 * h;
#pragma unroll
        for (int t = 0; t < ML_KS; ++t) xf[t] = *reinterpret_cast<const e8*>(xrow + 16 * t);
    }

    // ---- fragment addresses.  First product, k-step t: piece 2t + h of image row r, at
    // 16 (t >> 3) + ((2 (t & 7) + h) ^ (r & 15)); second product, k-step t of tile m: piece 2t + h of
    // image row 32 m + r.
    const unsigned char* a1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a1[u] = smem + r * (ML_HID * 2) + 16 * ((2 * u + h) ^ (r & 15));
    const unsigned char* a2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) a2[t] = smem + ML_OFF_W2 + r * (ML_C * 2) + 16 * ((2 * t + h) ^ ((r >> 2) & 3));
    // registers 8t..8t+7 of an H^T accumulator are units 16t + 8h + 0..7 of its chunk
    const unsigned char* b1s = smem + ML_OFF_B1 + 32 * h;
    const uint32_t l1[8] = {(uint32_t)(uintptr_t)a1[0], (uint32_t)(uintptr_t)a1[1], (uint32_t)(uintptr_t)a1[2],
                            (uint32_t)(uintptr_t)a1[3], (uint32_t)(uintptr_t)a1[4], (uint32_t)(uintptr_t)a1[5],
                            (uint32_t)(uintptr_t)a1[6], (uint32_t)(uintptr_t)a1[7]};
    const uint32_t l2[2] = {(uint32_t)(uintptr_t)a2[0], (uint32_t)(uintptr_t)a2[1]};
    // (a read's immediate offset has 16 bits: the third ring slot of the W2 images lies beyond them)
    const uint32_t l2c[2] = {l2[0] + 2 * ML_W_BYTES, l2[1] + 2 * ML_W_BYTES};
    const uint32_t lb = (uint32_t)(uintptr_t)b1s;

    auto gemm1 = [&](int buf, f32x16& acc) {                  // prologue only: compiler-scheduled reads
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int t = 0; t < ML_KS; ++t) {
            const e8 a = *reinterpret_cast<const e8*>(a1[t & 7] + buf * ML_W_BYTES + 256 * (t >> 3));
            acc = El16<E>::mfma(a, xf[t], acc);
        }
    };
    auto gelu_pack = [&](int c, const f32x16& acc, e8* pf) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const f32x2 b = *reinterpret_cast<const f32x2*>(b1s + c * (ML_C * 4) + 64 * (i >> 3) + 4 * (i & 7));
            const f32x2 g = gelu2((f32x2){acc[i], acc[i + 1]} + b);
            pf[i >> 3][i & 7] = (E)g.x;
            pf[i >> 3][(i & 7) + 1] = (E)g.y;
        }
    };

    f32x16 yacc[ML_YT];
#pragma unroll
    for (int m = 0; m < ML_YT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) yacc[m][i] = 0.f;

    if constexpr (PRE_LN) {
      if constexpr (PROJ) {
        // ---- attention-output projection and its LayerNorm:  X = LN1(x_in + ctx Wo^T + bo).
        // Y1^T[feature][token] = Wo . ctx^T into the same twelve accumulator tiles, Wo streamed through
        // the two W2 buffers in 12 slices of 32 input columns (the W2 image shape: 24 MFMAs per slice,
        // fragment reads issued ML_PO MFMAs ahead).  With the Wo rows in the swapped order, registers
        // 8t..8t+7 of tile m are features 32m + 16t + 8h + 0..7 of the lane's token: after the LayerNorm
        // they are, converted to the 16-bit element type, exactly the fragment xf[2m + t].
        constexpr int ML_PO = 6;
        auto dma_wo = [&](int kc, int buf, int i) {
            lds_dma16(lds0 + ML_OFF_W2 + buf * ML_W_BYTES + 1024 * (wave + 4 * i), p.wo + kc * ML_C, offo[i]);
        };
#pragma unroll
        for (int i = 0; i < ML_W_DMAS; ++i) dma_wo(0, 0, i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        auto slice = [&](auto kc_c) {
            constexpr int KC = decltype(kc_c)::value, BUF = KC & 1;
            u32x4 q[ML_PO + 1];
            auto issue = [&](auto s_c) {
                constexpr int S = decltype(s_c)::value;
                lds_read16<BUF * ML_W_BYTES + (S % ML_YT) * (32 * ML_C * 2)>(q[S % (ML_PO + 1)], l2[S / ML_YT]);
            };
            auto step = [&](auto s_c) {
                constexpr int S = decltype(s_c)::value;
                if constexpr (S < ML_W_DMAS && KC + 1 < ML_HID / ML_C) dma_wo(KC + 1, BUF ^ 1, S);
                if constexpr (S + ML_PO < ML_KS) issue(std::integral_constant<int, S + ML_PO>{});
                constexpr int newer = ML_KS - 1 - S < ML_PO ? ML_KS - 1 - S : ML_PO;
                lds_wait<newer>(q[S % (ML_PO + 1)]);
                constexpr int M = S % ML_YT, T = S / ML_YT;
                yacc[M] = El16<E>::mfma(__builtin_bit_cast(e8, q[S % (ML_PO + 1)]),
                                                                  xf[2 * KC + T], yacc[M]);
            };
            unrolled(std::make_integer_sequence<int, ML_PO>{}, issue);
            unrolled(std::make_integer_sequence<int, ML_KS>{}, step);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        };
        unrolled(std::make_integer_sequence<int, ML_HID / ML_C>{}, slice);
        dma_w2(0, 0);                                         // the W2 buffers are free again
        dma_w2(min(1, last), 1);
      }
        // bias, residual, LayerNorm; the result replaces the ctx fragments in xf
        // (MODE 2: the accumulator registers only serve as the fp32 copy of the row)
        if constexpr (!PROJ) __syncthreads();                 // gamma1 | beta1 were staged by other threads
        const unsigned char* v1 = smem + ML_OFF_B1 + 4 * (size_t)p.I + 2 * (3 * ML_HID) + 16 * h;   // bo | gamma1 | beta1
        const E* xin = p.x + tok_ld * ML_HID + 8 * h;
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < ML_KS; ++t) {
            const e8 xr = *reinterpret_cast<const e8*>(xin + 16 * t);
            const e8 bb = *reinterpret_cast<const e8*>(v1 + 32 * t);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = PROJ ? yacc[t >> 1][8 * (t & 1) + j] + (float)bb[j] + (float)xr[j]
                                     : (float)xf[t][j] + (float)xr[j];
                yacc[t >> 1][8 * (t & 1) + j] = v;
                sum += v;
            }
        }
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum / (float)ML_HID;
        float sq = 0.f;
#pragma unroll
        for (int m = 0; m < ML_YT; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) { const float d = yacc[m][i] - mean; sq += d * d; }
        sq += __shfl_xor(sq, 32, 64);
        const float rstd = rsqrtf(sq / (float)ML_HID + p.eps);
#pragma unroll
        for (int t = 0; t < ML_KS; ++t) {
            const e8 g8 = *reinterpret_cast<const e8*>(v1 + 2 * ML_HID + 32 * t);
            const e8 be8 = *reinterpret_cast<const e8*>(v1 + 4 * ML_HID + 32 * t);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xf[t][j] = (E)((yacc[t >> 1][8 * (t & 1) + j] - mean) * rstd * (float)g8[j] + (float)be8[j]);
                yacc[t >> 1][8 * (t & 1) + j] = 0.f;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- prologue: H^T of chunks 0 and 1; then W1 buffer 0 is free for chunk 3
    f32x16 hA, hB;
    e8 pA[2], pB[2];
    gemm1(0, hA);
    gemm1(1, hB);
    __syncthreads();
    dma_w1(min(3, last), 0);
    gelu_pack(0, hA, pA);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- main loop, three chunks in flight: iteration c runs H^T of chunk c+2 (into hnew), the GELU
    // of chunk c+1 (hold -> pnext) and Y^T += W2 . H^T of chunk c (pcur) — independent of one another,
    // so the matrix unit, the vector ALU and the LDS reads overlap — while chunks c+4 of W1 and c+2 of
    // W2 stream into the ring slots the previous iteration finished with.  The weights an iteration asks
    // for are used by the iteration after the next: the wait that closes an iteration leaves its own loads
    // in flight (counted vmcnt) and only makes sure of the previous iteration's, so a chunk's 48 KiB have a
    // whole iteration to pass the CU's address path instead of the half iteration left behind their issue
    // (waves were parked at that wait 28 % of their life).  Past the last chunk the loads and the first
    // product repeat the last chunk (results unused): no branches in the body.  Six iterations are written
    // out — ring slot (three) x roles of the accumulators (two) — with the slot a constant, so every
    // fragment address is a fixed register plus an immediate.
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#if ML_STAMPS
    unsigned long long st[5] = {0, 0, 0, 0, 0};
#endif
    auto iteration = [&](int c, auto buf_c, f32x16& hnew, const f32x16& hold, const e8* pcur, e8* pnext) {
        constexpr int R = decltype(buf_c)::value;          // c % 3: W2 chunk c sits in slot R, W1 chunk c + 2 in slot (R + 2) % 3
        constexpr int RB1 = (R + 2) % ML_RING, WB1 = (R + 1) % ML_RING, WB2 = (R + 2) % ML_RING;
        ML_STAMP(t0);
        // the weight loads are issued one per k-step over the first half of the steps, between MFMAs
        // (issuing one costs the wave some 50 cycles, which the matrix unit covers there and would not
        // cover in one block up front; the second half of the steps gives the last ones time to land)
        const E* g1 = p.w1 + (size_t)min(c + 4, last) * ML_C * ML_HID;
        const E* g2 = p.w2 + min(c + 2, last) * ML_C;
        ML_STAMP(t1);
        // bias of chunk c+1 (for the GELU), then the fragment ring: k-step s uses slot s % (ML_PF + 1)
        f32x4 bias[4];
        const uint32_t lbc = lb + (uint32_t)min(c + 1, last) * (ML_C * 4);
        lds_read16<0>(bias[0], lbc);
        lds_read16<16>(bias[1], lbc);
        lds_read16<64>(bias[2], lbc);
        lds_read16<80>(bias[3], lbc);
        u32x4 q1[ML_PF + 1], q2[ML_PF + 1];
        f32x2 gv, gc, gs, gp;
        auto issue = [&](auto s_c) {
            constexpr int S = decltype(s_c)::value;
            lds_read16<RB1 * ML_W_BYTES + 256 * (S >> 3)>(q1[S % (ML_PF + 1)], l1[S & 7]);
            if constexpr (R == 2) lds_read16<(S % ML_YT) * (32 * ML_C * 2)>(q2[S % (ML_PF + 1)], l2c[S / ML_YT]);
            else lds_read16<R * ML_W_BYTES + (S % ML_YT) * (32 * ML_C * 2)>(q2[S % (ML_PF + 1)], l2[S / ML_YT]);
        };
        auto step = [&](auto s_c) {
            constexpr int S = decltype(s_c)::value;
            if constexpr (S % ML_DMA_EVERY == 0 && S / ML_DMA_EVERY < 2 * ML_W_DMAS && !ML_ABLATE_DMA) {
                constexpr int G = S / ML_DMA_EVERY;
                if constexpr (G < ML_W_DMAS)
                    lds_dma16(lds0 + WB1 * ML_W_BYTES + 1024 * (wave + 4 * G), g1, off1[G]);
                else
                    lds_dma16(lds0 + ML_OFF_W2 + WB2 * ML_W_BYTES + 1024 * (wave + 4 * (G - ML_W_DMAS)), g2,
                              off2[G - ML_W_DMAS]);
            }
            if constexpr (S + ML_PF < ML_KS) issue(std::integral_constant<int, S + ML_PF>{});
            constexpr int newer = 2 * (ML_KS - 1 - S < ML_PF ? ML_KS - 1 - S : ML_PF);   // reads issued after this step's
            lds_wait<newer>(q1[S % (ML_PF + 1)], q2[S % (ML_PF + 1)]);
            if constexpr (S == 0)
                hnew = El16<E>::mfma(__builtin_bit_cast(e8, q1[0]), xf[0], zero16);
            else
                hnew = El16<E>::mfma(__builtin_bit_cast(e8, q1[S % (ML_PF + 1)]), xf[S], hnew);
            constexpr int M = S % ML_YT, T = S / ML_YT;
            yacc[M] = El16<E>::mfma(__builtin_bit_cast(e8, q2[S % (ML_PF + 1)]), pcur[T], yacc[M]);
            // the GELU of one register pair is spread over three k-steps, each slice pinned behind this
            // step's MFMAs (the empty asm statements keep their place among the LDS statements), so the
            // vector work issues while the matrix unit is busy instead of in one block.  (Pinning the
            // MFMAs themselves the same way costs more than it gains: the compiler then pads each with
            // the wait states of a reader of its result.)
            constexpr int I = 2 * (S / 3);                     // registers I, I+1 of the accumulator
            if constexpr (S % 3 == 0) {
                const f32x4 b4 = bias[I >> 2];
                const f32x2 b = (I & 2) ? (f32x2){b4[2], b4[3]} : (f32x2){b4[0], b4[1]};
                gv = (f32x2){hold[I], hold[I + 1]} + b;
                gelu_head(gv, gc, gs, gp);
                asm volatile("" : "+v"(gp), "+v"(gs));
            } else if constexpr (S % 3 == 1) {
                gelu_mid(gs, gp);
                asm volatile("" : "+v"(gp));
            } else {
                f32x2 g = gelu_tail(gv, gc, gs, gp);
                asm volatile("" : "+v"(g));
                pnext[I >> 3][I & 7] = (E)g.x;
                pnext[I >> 3][(I & 7) + 1] = (E)g.y;
            }
        };
        unrolled(std::make_integer_sequence<int, ML_PF>{}, issue);
        // the bias reads were issued before the first 2 ML_PF fragment reads
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2]), "+v"(bias[3]) : "n"(2 * ML_PF));
        unrolled(std::make_integer_sequence<int, ML_KS>{}, step);
        ML_STAMP(t2);
        // this iteration's own 2 * ML_W_DMAS loads may still be in flight; everything older has landed
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ML_W_DMAS) : "memory");
        ML_STAMP(t3);
        __syncthreads();
        ML_STAMP(t4);
#if ML_STAMPS
        st[0] += t1 - t0; st[1] += t2 - t1; st[2] += t3 - t2; st[3] += t4 - t3; st[4] += 1;
#endif
    };
#if ML_STAMPS
    ML_STAMP(tl0);
#endif
    for (int c = 0; c < n_chunks; c += 6) {
        iteration(c, std::integral_constant<int, 0>{}, hA, hB, pA, pB);
        if (c + 1 < n_chunks) iteration(c + 1, std::integral_constant<int, 1>{}, hB, hA, pB, pA);
        if (c + 2 < n_chunks) iteration(c + 2, std::integral_constant<int, 2>{}, hA, hB, pA, pB);
        if (c + 3 < n_chunks) iteration(c + 3, std::integral_constant<int, 0>{}, hB, hA, pB, pA);
        if (c + 4 < n_chunks) iteration(c + 4, std::integral_constant<int, 1>{}, hA, hB, pA, pB);
        if (c + 5 < n_chunks) iteration(c + 5, std::integral_constant<int, 2>{}, hB, hA, pB, pA);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the repeats of the last chunk: nothing may land in LDS behind this point

#if ML_STAMPS
    ML_STAMP(tl1);
    const unsigned long long tpro = tl0 - tk0;
#endif
    // ---- epilogue.  The W2 image rows are in the same swapped order as the W1 rows, so registers
    // 8t'..8t'+7 of tile m are features 32m + 16t' + 8h + 0..7 of the lane's token: the layout of the X
    // fragments.  The residual (the block's own input) is therefore xf[2m + t'] itself, and bias, gamma,
    // beta (LDS) and the output are 16-byte accesses.
    const unsigned char* vecs = smem + ML_OFF_B1 + 4 * (size_t)p.I + 16 * h;
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < ML_KS; ++t) {
        const e8 bb = *reinterpret_cast<const e8*>(vecs + 32 * t);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = yacc[t >> 1][8 * (t & 1) + j] + (float)bb[j] + (float)xf[t][j];
            yacc[t >> 1][8 * (t & 1) + j] = v;
            sum += v;
        }
    }
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum / (float)ML_HID;
    float sq = 0.f;
#pragma unroll
    for (int m = 0; m < ML_YT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float d = yacc[m][i] - mean; sq += d * d; }
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = rsqrtf(sq / (float)ML_HID + p.eps);
    const bool live = tok < p.M;
#pragma unroll
    for (int t = 0; t < ML_KS; ++t) {
        const e8 g8 = *reinterpret_cast<const e8*>(vecs + 2 * ML_HID + 32 * t);
        const e8 be8 = *reinterpret_cast<const e8*>(vecs + 4 * ML_HID + 32 * t);
        e8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            o[j] = (E)((yacc[t >> 1][8 * (t & 1) + j] - mean) * rstd * (float)g8[j] + (float)be8[j]);
        if (live) *reinterpret_cast<e8*>(p.y + tok * ML_HID + 16 * t + 8 * h) = o;
    }
#if ML_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ML_STAMP(tk1);
    if (blockIdx.x == 300 && tid == 64) { for (int i = 0; i < 5; ++i) ml_dbg[i] = st[i]; ml_dbg[5] = tl1 - tl0; ml_dbg[6] = tpro; ml_dbg[7] = tk1 - tl1; }
#endif
}


// ---- Y = X W^T + bias for 384-wide inputs (the q|k|v projection) ------------------------------------
// The first product of the block above, on its own: a wave owns 32 tokens whose X fragments stay in
// registers, W streams through LDS in images of 32 output features (the W1 image: swizzled, rows in the
// swapped order, so registers 8t..8t+7 of the accumulator are features 16t + 8h + 0..7 and the output
// is stored in 16-byte pieces).  24 MFMAs per image alternate between two accumulators.  This kernel
// needs few registers, so TWO workgroups share a CU and the second wave of a SIMD fills the barrier,
// the weight loads and the stores of the other.  The stores of an image are issued during the
// next one, BEHIND that image's weight loads: loads and stores retire in issue order, so the counted wait
// that closes the image (all but the two newest operations) retires the weight loads and leaves the
// stores in flight for another image (waiting for them there cost 5 % of the kernel).
// Measured with the stores removed the q|k|v shape runs 32 % faster, with the weight loads removed 15 %:
// the 2.3 KB a token this kernel writes, in 32-byte pieces, is what bounds it.
#ifndef LN_PF_N
#define LN_PF_N 4
#endif
constexpr int LN_PF = LN_PF_N;
#ifndef LN_ABLATE_STORE
#define LN_ABLATE_STORE 0          // measurement builds: results kept alive but not stored
#endif
#ifndef LN_PAIR_STORE
#define LN_PAIR_STORE 1            // 0: every lane stores its own two 16-byte pieces (32 contiguous bytes per row)
#endif
#ifndef LN_LATE_STORE
#define LN_LATE_STORE 1            // 0: stores at the start of the next image, full wait at its end
#endif
#ifndef LN_STAMPS
#define LN_STAMPS 0                // measurement builds: cycle stamps of one wave (rag_enc_linear_debug)
#endif
#if LN_STAMPS
__device__ unsigned long long ln_dbg[8];
#define LN_STAMP(var) unsigned long long var = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define LN_STAMP(var)
#endif
#ifndef LN_ABLATE_LDS
#define LN_ABLATE_LDS 0            // measurement builds: only the first fragments of an image are read
#endif
#ifndef LN_ABLATE_BARRIER
#define LN_ABLATE_BARRIER 0        // measurement builds: no barrier between images
#endif
#ifndef LN_ABLATE_DMA
#define LN_ABLATE_DMA 0            // measurement builds: the images after the first are not loaded
#endif

template <typename E>
struct LinearParamsT {
    const E* x;        // [M][384]
    const E* w;        // [N][384]
    const E* bias;     // [N]
    E* y;              // [M][N]
    int64_t M;
    int N;
    // LN kernels (N == 384): y = LayerNorm(x W^T + bias + residual) * gamma + beta
    const E* residual; // [M][384]
    const E* gamma;    // [384]
    const E* beta;     // [384]
    float eps;
};

// LN: the sum is rounded to bf16 and stored image by image while the row statistics are gathered
// from the rounded values; a second pass over the wave's own rows (just written, still in cache)
// normalises them in place.
template <typename E, bool LN>
__global__ __launch_bounds__(ML_NT, 2) void linear384_kernel(LinearParamsT<E> p) {
    typedef typename El16<E>::x8 e8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // W images [2][32][48 x 16 B], bias fp32 [N]
    LN_STAMP(tk0);
#if LN_STAMPS
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int64_t tok = (int64_t)blockIdx.x * ML_BM + wave * 32 + r;
    const int64_t tok_ld = min(tok, p.M - 1);
    const bool live = tok < p.M;
    // uniform: the wave issues both of an image's stores (with a single live row only one of the pair form's)
    const bool wave_live = (int64_t)blockIdx.x * ML_BM + wave * 32 + (LN_PAIR_STORE ? 1 : 0) < p.M;
    const int n_img = p.N / ML_C;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

    uint32_t off1[ML_W_DMAS];
#pragma unroll
    for (int i = 0; i < ML_W_DMAS; ++i) {
        const int L = 64 * (wave + 4 * i) + lane;
        const int q = L / 48, j1 = (L % 48) ^ (q & 15);
        const int unit = (q & ~12) | ((q & 4) << 1) | ((q & 8) >> 1);
        off1[i] = (uint32_t)((unit * ML_HID + j1 * 8) * 2);
    }
    auto dma = [&](int c, int buf, int i) {
        lds_dma16(lds0 + buf * ML_W_BYTES + 1024 * (wave + 4 * i), p.w + (size_t)c * ML_C * ML_HID, off1[i]);
    };
#pragma unroll
    for (int i = 0; i < ML_W_DMAS; ++i) dma(0, 0, i);
    for (int i = tid; i < p.N; i += ML_NT) *reinterpret_cast<float*>(smem + 2 * ML_W_BYTES + 4 * i) = (float)p.bias[i];

    e8 xf[ML_KS];
    {
        const E* xrow = p.x + tok_ld * ML_HID + 8 * h;
#pragma unroll
        for (int t = 0; t < ML_KS; ++t) xf[t] = *reinterpret_cast<const e8*>(xrow + 16 * t);
    }
    uint32_t l1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) l1[u] = lds0 + r * (ML_HID * 2) + 16 * ((2 * u + h) ^ (r & 15));
    const unsigned char* bias_s = smem + 2 * ML_W_BYTES + 32 * h;
    E* yrow = p.y + tok * p.N + 8 * h;
    const E* rrow = LN ? p.residual + tok_ld * ML_HID + 8 * h : nullptr;
    float rsum = 0.f, rsq = 0.f;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    e8 out0, out1;                                        // the previous image's results, stored one image later
#if LN_STAMPS
    LN_STAMP(tp);
    unsigned long long st[4] = {0, 0, 0, 0};
#endif
    auto image = [&](int c, auto buf_c) {
        constexpr int BUF = decltype(buf_c)::value;
        LN_STAMP(t0);
        auto store_prev = [&]() {
            if (LN_ABLATE_STORE) {
                asm volatile("" ::"v"(out0), "v"(out1));
            } else if (!LN && LN_PAIR_STORE) {
                // A lane holds pieces h and 2 + h (16 bytes each) of its row's 64 bytes.  The lanes of two
                // neighbouring rows swap one piece each: the even row's lanes send piece 2 + h and get the
                // odd row's piece h.  Store A then writes the even row from all four lanes, store B the odd
                // row: 64 contiguous bytes per row and instruction instead of 32.
                if (c > 0) {
                    const bool odd = r & 1;
                    const u32x4 a = __builtin_bit_cast(u32x4, out0), b = __builtin_bit_cast(u32x4, out1);
                    u32x4 got;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        got[k] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(odd ? a[k] : b[k]), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                    const u32x4 va = odd ? got : a, vb = odd ? b : got;
                    E* base = p.y + (tok & ~(int64_t)1) * p.N + (c - 1) * ML_C + (odd ? 16 : 0) + 8 * h;
                    if ((tok & ~(int64_t)1) < p.M) *reinterpret_cast<u32x4*>(base) = va;
                    if ((tok | 1) < p.M) *reinterpret_cast<u32x4*>(base + p.N) = vb;
                }
            } else if (c > 0 && live) {
                *reinterpret_cast<e8*>(yrow + (c - 1) * ML_C) = out0;
                *reinterpret_cast<e8*>(yrow + (c - 1) * ML_C + 16) = out1;
            }
        };
        if constexpr (LN || !LN_LATE_STORE) store_prev();
        const int cn = min(c + 1, n_img - 1);
        e8 res0, res1;
        if constexpr (LN) {
            res0 = *reinterpret_cast<const e8*>(rrow + c * ML_C);
            res1 = *reinterpret_cast<const e8*>(rrow + c * ML_C + 16);
        }
        u32x4 q[LN_PF + 1];
        f32x16 ae, ao;
        auto issue = [&](auto s_c) {
            constexpr int S = decltype(s_c)::value;
            if constexpr (!LN_ABLATE_LDS || S <= LN_PF)
                lds_read16<BUF * ML_W_BYTES + 256 * (S >> 3)>(q[S % (LN_PF + 1)], l1[S & 7]);
        };
        auto step = [&](auto s_c) {
            constexpr int S = decltype(s_c)::value;
            if constexpr (S < ML_W_DMAS && !LN_ABLATE_DMA) dma(cn, BUF ^ 1, S);
            if constexpr (!LN && LN_LATE_STORE && S == ML_W_DMAS) store_prev();
            if constexpr (S + LN_PF < ML_KS) issue(std::integral_constant<int, S + LN_PF>{});
            constexpr int newer = ML_KS - 1 - S < LN_PF ? ML_KS - 1 - S : LN_PF;
            lds_wait<newer>(q[S % (LN_PF + 1)]);
            const e8 a = __builtin_bit_cast(e8, q[S % (LN_PF + 1)]);
            if constexpr (S == 0)
                ae = El16<E>::mfma(a, xf[0], zero16);
            else if constexpr (S == 1)
                ao = El16<E>::mfma(a, xf[1], zero16);
            else if constexpr (S % 2 == 0)
                ae = El16<E>::mfma(a, xf[S], ae);
            else
                ao = El16<E>::mfma(a, xf[S], ao);
        };
        unrolled(std::make_integer_sequence<int, LN_PF>{}, issue);
        unrolled(std::make_integer_sequence<int, ML_KS>{}, step);
        LN_STAMP(t1);
        // registers 8t..8t+7 are features 32c + 16t + 8h + 0..7
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_s + c * (ML_C * 4));
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias_s + c * (ML_C * 4) + 16);
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(bias_s + c * (ML_C * 4) + 64);
        const f32x4 b3 = *reinterpret_cast<const f32x4*>(bias_s + c * (ML_C * 4) + 80);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v0 = ae[j] + ao[j] + (j < 4 ? b0[j & 3] : b1[j & 3]);
            float v1 = ae[8 + j] + ao[8 + j] + (j < 4 ? b2[j & 3] : b3[j & 3]);
            if constexpr (LN) {
                v0 += (float)res0[j];
                v1 += (float)res1[j];
            }
            out0[j] = (E)v0;
            out1[j] = (E)v1;
            if constexpr (LN) {
                const float r0 = (float)out0[j], r1 = (float)out1[j];
                rsum += r0 + r1;
                rsq += r0 * r0 + r1 * r1;
            }
        }
        LN_STAMP(t2);
        if constexpr (!LN && LN_LATE_STORE && !LN_ABLATE_STORE) {
            // the two stores were issued behind the image's loads and may stay in flight: loads and stores
            // retire in issue order, so two outstanding operations mean every weight load has landed
            if (c > 0 && wave_live) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!LN_ABLATE_BARRIER) __builtin_amdgcn_s_barrier();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!LN_ABLATE_BARRIER) __syncthreads();
        }
#if LN_STAMPS
        LN_STAMP(t3);
        st[0] += t1 - t0; st[1] += t2 - t1; st[2] += t3 - t2; st[3] += 1;
#endif
    };
    for (int c = 0; c < n_img; c += 2) {
        image(c, std::integral_constant<int, 0>{});
        if (c + 1 < n_img) image(c + 1, std::integral_constant<int, 1>{});
    }
    if (live) {
        *reinterpret_cast<e8*>(yrow + (n_img - 1) * ML_C) = out0;
        *reinterpret_cast<e8*>(yrow + (n_img - 1) * ML_C + 16) = out1;
    }
#if LN_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LN_STAMP(tk1);
    if (blockIdx.x == 300 && tid == 64) { ln_dbg[0] = st[0]; ln_dbg[1] = st[1]; ln_dbg[2] = st[2]; ln_dbg[3] = st[3]; ln_dbg[4] = tp - tk0; ln_dbg[5] = tk1 - tp; ln_dbg[6] = tk1 - tk0; ln_dbg[7] = __builtin_amdgcn_s_memrealtime() - rt0; }
#endif
    if constexpr (LN) {
        // a token's 384 values are split between this lane and its partner in the other half-wave
        rsum += __shfl_xor(rsum, 32, 64);
        rsq += __shfl_xor(rsq, 32, 64);
        const float mean = rsum / (float)ML_HID;
        const float var = fmaxf(rsq / (float)ML_HID - mean * mean, 0.f);
        const float rstd = rsqrtf(var + p.eps);
        if (live) {
#pragma unroll 4
            for (int t = 0; t < ML_KS; ++t) {
                const e8 v = *reinterpret_cast<const e8*>(yrow + 16 * t);
                const e8 g8 = *reinterpret_cast<const e8*>(p.gamma + 16 * t + 8 * h);
                const e8 be8 = *reinterpret_cast<const e8*>(p.beta + 16 * t + 8 * h);
                e8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (E)(((float)v[j] - mean) * rstd * (float)g8[j] + (float)be8[j]);
                *reinterpret_cast<e8*>(yrow + 16 * t) = o;
            }
        }
    }
}



// ---- q|k|v projection and self-attention of whole sequences in one kernel ---------------------------
// The projection above writes 2304 bytes a token that the attention kernel reads back once; here they
// never leave the chip.  A workgroup of eight waves takes a TILE: up to 256 consecutive tokens of a
// packed batch that are whole sequences (the host cuts the batch into tiles).  A wave owns 32 tokens,
// their X fragments in registers as in linear384_kernel; the weight images stream through a ring of three
// LDS buffers in the order q, k, v of head 0, q, k, v of head 1, ... (so q images always sit in buffer 0, k in 1,
// v in 2), each asked for two images before its use (one workgroup per CU: nothing else would cover the wait):
//   q image: the accumulators (features in registers, the wave's tokens in lanes) are, rounded to the
//            16-bit type, exactly the B fragments Q^T of the score product — they stay in registers;
//   k image: the same layout, written as rows of the K image in LDS;
//   v image: computed with the operands swapped (tokens in registers, features in lanes), which is the
//            orientation of the V^T image: each lane writes four tokens of its feature row at a time.
// Behind the barrier that closes the v image every wave runs the attention of its 32 queries over the
// key blocks that hold its queries' sequences (S^T = K Q^T, online softmax in base 2, O^T += V^T P^T, as
// attention_kernel in encoder.hip; two key blocks per step; keys of other sequences masked by a code product on
// the matrix unit, see below).  The K and V^T
// images are overwritten by the next head's k and v images, which every wave reaches only through the
// barrier that closes the next q image, i.e. after all waves have finished this head's attention.
// HBM traffic: 768 B read and 768 B written per token.
constexpr int QA_WAVES = 8;
constexpr int QA_NT = 64 * QA_WAVES;
constexpr int QA_TILE = 32 * QA_WAVES;                     // tokens of a tile
constexpr int QA_RING = 3;                                 // weight buffers
constexpr int QA_DPW = ML_W_BYTES / 1024 / QA_WAVES;      // weight loads per wave and image: 3
constexpr int QA_D = 32;                                   // head size
constexpr int QA_KROW = QA_D + 8;                          // K rows padded by 16 bytes
constexpr int QA_VROW = QA_TILE + 4;                       // V^T rows padded by 8 bytes
constexpr int QA_OFF_BIAS = QA_RING * ML_W_BYTES;
constexpr int QA_OFF_K = QA_OFF_BIAS + 4 * 3 * ML_HID;
constexpr int QA_OFF_V = QA_OFF_K + QA_TILE * QA_KROW * 2;
constexpr int QA_OFF_KC = QA_OFF_V + QA_D * QA_VROW * 2;   // sequence codes of the tile's tokens, [256][16]
constexpr int QA_OFF_SEQ = QA_OFF_KC + QA_TILE * 16 * 2;   // tile-local start of every sequence of the tile, [257] int32
constexpr int QA_LDS = QA_OFF_SEQ + (QA_TILE + 4) * 4;     // 124688 bytes: one workgroup per CU
constexpr float QA_CODE = 2048.f;                          // weight of a code bit in the score
// (A four-wave form with two weight buffers and 128-token tiles, two workgroups per CU whose phases interleave, ran
// 6 % faster on sequences of up to 128 tokens — 1.43 against 1.52 ms per 1M tokens — and would need the eight-wave
// form beside it for longer ones; not kept.)
constexpr float QA_NEG = -1e30f;

#ifndef QA_STAMPS
#define QA_STAMPS 0                // measurement builds: cycle stamps of one wave (rag_enc_qkv_attention_debug)
#endif
#if QA_STAMPS
__device__ unsigned long long qa_dbg[8];
#define QA_STAMP(var) unsigned long long var = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define QA_STAMP(var)
#endif

template <typename E>
struct QkvAttnParams {
    const E* x;              // [T][384] layer input
    const E* w;              // [1152][384] q | k | v projection weights, heads inside each
    const E* bias;           // [1152]
    E* out;                  // [T][384] attention output
    const int32_t* cu;       // [n_seq + 1] token offsets of the packed sequences
    const int32_t* tile_seq; // [n_tiles + 1] first sequence of every tile
    int64_t T;
    int n_seq;
    float scale;             // log2(e) / sqrt(head size)
    uint32_t* fault;
};

// FIRST: only the first token of every sequence is a query ([CLS] of the last layer, the one row the pooler reads);
// out is then [n_seq][384], row = sequence.
template <typename E, bool FIRST>
__global__ __launch_bounds__(QA_NT, 1) void qkv_attention_kernel(QkvAttnParams<E> p) {
    typedef typename El16<E>::x8 e8;
    typedef E e4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    QA_STAMP(tk0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

    // ---- the tile: sequences [sa, sb), tokens [T0, T1)
    const int sa = p.tile_seq[blockIdx.x], sb = p.tile_seq[blockIdx.x + 1];
    if (sa < 0 || sb <= sa || sb > p.n_seq || sb - sa > QA_TILE) {      // (a sequence has at least one token)
        if (tid == 0) rag_flag(p.fault, RAG_FAULT_SEGMENT);
        return;
    }
    const int64_t T0 = p.cu[sa], T1 = p.cu[sb];
    if (T0 < 0 || T1 <= T0 || T1 - T0 > QA_TILE || T1 > p.T) {        // not a tile of this batch
        if (tid == 0) rag_flag(p.fault, RAG_FAULT_SEGMENT);
        return;
    }
    const int ntok = (int)(T1 - T0);
    const int ltok = wave * 32 + r;                          // this lane's token inside the tile
    const bool valid = ltok < ntok;
    const int64_t tok_ld = T0 + min(ltok, ntok - 1);         // lanes past the tile read its last row; never stored

    // the X rows first: everything below up to the barrier runs under their flight
    e8 xf[ML_KS];
    {
        const E* xrow = p.x + tok_ld * ML_HID + 8 * h;
#pragma unroll
        for (int t = 0; t < ML_KS; ++t) xf[t] = *reinterpret_cast<const e8*>(xrow + 16 * t);
    }
    uint32_t off1[QA_DPW];
#pragma unroll
    for (int i = 0; i < QA_DPW; ++i) {
        const int L = 64 * (wave + QA_WAVES * i) + lane;
        const int q = L / 48, j1 = (L % 48) ^ (q & 15);
        const int unit = (q & ~12) | ((q & 4) << 1) | ((q & 8) >> 1);
        off1[i] = (uint32_t)((unit * ML_HID + j1 * 8) * 2);
    }
    // image c = 3 * head + (0: q, 1: k, 2: v) holds weight rows [which * 384 + 32 * head, + 32)
    auto dma = [&](int c, int buf, int i) {
        const int head = c / 3, which = c - 3 * head;
        lds_dma16(lds0 + buf * ML_W_BYTES + 1024 * (wave + QA_WAVES * i),
                  p.w + (size_t)(which * ML_HID + QA_D * head) * ML_HID, off1[i]);
    };
#pragma unroll
    for (int i = 0; i < QA_DPW; ++i) dma(0, 0, i);
#pragma unroll
    for (int i = 0; i < QA_DPW; ++i) dma(1, 1, i);
    for (int i = tid; i < 3 * ML_HID; i += QA_NT) *reinterpret_cast<float*>(smem + QA_OFF_BIAS + 4 * i) = (float)p.bias[i];

    // this lane's sequence, as tile-local token range [s0, s1)
    // (the starts of the tile's sequences go through LDS — one load per thread — and every lane finds its own by
    // bisection: a loop over the sequences with a scalar load each costs a memory round trip per sequence)
    int s0 = 0, s1 = 0, sid = 0;                              // sid: 1 + index of the sequence in the tile, 0 = no token
    {
        int* seq_start = reinterpret_cast<int*>(smem + QA_OFF_SEQ);
        const int n_s = sb - sa;                             // <= 256: every sequence has a token
        for (int i = tid; i <= n_s; i += QA_NT) seq_start[i] = (int)(p.cu[sa + i] - T0);
        __syncthreads();
        int lo = 0, hi = n_s;                                // the sequence j with seq_start[j] <= ltok < seq_start[j + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (seq_start[mid] <= ltok) lo = mid; else hi = mid;
        }
        if (valid) { s0 = seq_start[lo]; s1 = seq_start[lo + 1]; sid = lo + 1; }
    }
    // Keys of other sequences are masked BY THE MATRIX UNIT: every token carries a 16-wide code — bits 0..8 of sid
    // as +-1, then a 1 — and every query the same bits times QA_CODE, then -9 QA_CODE.  The product of a key's and a
    // query's code is 0 for the same sequence and at most -2 QA_CODE otherwise (all terms are integers: exact), and
    // is taken as a third k-step of the score product, first, into the zero accumulator.  A masked score then sits
    // 4096 below any real one: its weight 2^(-4096 scale) is exactly 0.  (16 compare-and-select pairs per key block
    // and lane otherwise.)
    e8 qc;
    {
        e8 kc;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int bit = 8 * h + j;
            const float sgn = ((sid >> bit) & 1) ? 1.f : -1.f;
            kc[j] = (E)(bit < 9 ? sgn : bit == 9 ? 1.f : 0.f);
            qc[j] = (E)(bit < 9 ? sgn * QA_CODE : bit == 9 ? -9.f * QA_CODE : 0.f);
        }
        *reinterpret_cast<e8*>(smem + QA_OFF_KC + (ltok * 16 + 8 * h) * 2) = kc;
    }
    // key blocks this wave's queries need (uniform): from the first lane's sequence start to the last valid lane's end
    int n_mine = min(32, ntok - wave * 32);                  // valid lanes of this wave (<= 0: none)
    int kb_lo = 0, kb_hi = 0;
    const bool stores = valid && (!FIRST || ltok == s0);     // this lane's query is wanted
    if constexpr (FIRST) {
        // only the sequences that START in this wave's 32 tokens matter: from the first such lane's start to the last one's end
        const uint32_t starts = (uint32_t)__builtin_amdgcn_ballot_w64(stores);      // lanes 0..31 (the upper half repeats them)
        if (starts == 0) {
            n_mine = 0;                                      // no query here: the wave only projects
        } else {
            kb_lo = __builtin_amdgcn_readlane(s0, __builtin_ctz(starts)) & ~31;
            kb_hi = __builtin_amdgcn_readlane(s1, 31 - __builtin_clz(starts));
        }
    } else if (n_mine > 0) {
        kb_lo = __builtin_amdgcn_readfirstlane(s0) & ~31;
        kb_hi = __builtin_amdgcn_readlane(s1, n_mine - 1);
    }

    uint32_t l1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) l1[u] = lds0 + r * (ML_HID * 2) + 16 * ((2 * u + h) ^ (r & 15));
    const unsigned char* bias_s = smem + QA_OFF_BIAS;
    E* Ks = reinterpret_cast<E*>(smem + QA_OFF_K);
    E* Vt = reinterpret_cast<E*>(smem + QA_OFF_V);
    const int dswap = (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1);   // the feature that image row r holds
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    e8 qf[2];
#if QA_STAMPS
    QA_STAMP(tp);
    unsigned long long st[4] = {0, 0, 0, 0};
#endif
    // one weight image: 24 MFMAs over the X fragments; WHICH = 0 q, 1 k, 2 v
    auto image = [&](int head, auto which_c) {
        // image c = 3 head + WHICH sits in buffer c % 3 = WHICH; the image two ahead is loaded meanwhile
        constexpr int WHICH = decltype(which_c)::value, BUF = WHICH, NBUF = (BUF + 2) % QA_RING;
        const int c = 3 * head + WHICH;
        const int cn = min(c + 2, 3 * (ML_HID / QA_D) - 1);    // past the end: the last image again, unused
        u32x4 q[LN_PF + 1];
        f32x16 ae;
        QA_STAMP(t0);
        auto issue = [&](auto s_c) {
            constexpr int S = decltype(s_c)::value;
            lds_read16<BUF * ML_W_BYTES + 256 * (S >> 3)>(q[S % (LN_PF + 1)], l1[S & 7]);
        };
        auto step = [&](auto s_c) {
            constexpr int S = decltype(s_c)::value;
            if constexpr (S < QA_DPW) dma(cn, NBUF, S);
            if constexpr (S + LN_PF < ML_KS) issue(std::integral_constant<int, S + LN_PF>{});
            constexpr int newer = ML_KS - 1 - S < LN_PF ? ML_KS - 1 - S : LN_PF;
            lds_wait<newer>(q[S % (LN_PF + 1)]);
            const e8 a = __builtin_bit_cast(e8, q[S % (LN_PF + 1)]);
            // v image: tokens as rows (X is the A operand), features as columns.  One accumulator: an MFMA that
            // continues the accumulator of the one before it issues at the full rate.
            const f32x16 prev = S == 0 ? zero16 : ae;
            if constexpr (WHICH == 2) ae = El16<E>::mfma(xf[S], a, prev);
            else ae = El16<E>::mfma(a, xf[S], prev);
        };
        unrolled(std::make_integer_sequence<int, LN_PF>{}, issue);
        unrolled(std::make_integer_sequence<int, ML_KS>{}, step);
        QA_STAMP(t1);
        if constexpr (WHICH == 2) {
            // register i is token 8 (i >> 2) + 4 h + (i & 3) of the wave, column = feature dswap
            const float bv = *reinterpret_cast<const float*>(bias_s + 4 * (2 * ML_HID + QA_D * head + dswap));
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                e4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (E)(ae[4 * g + j] + bv);
                *reinterpret_cast<e4*>(Vt + dswap * QA_VROW + wave * 32 + 8 * g + 4 * h) = v;
            }
        } else {
            // registers 8t..8t+7 are features 16t + 8h + 0..7 of the lane's token
            const unsigned char* bp = bias_s + 4 * (WHICH * ML_HID + QA_D * head) + 32 * h;
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(bp + 16);
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(bp + 64);
            const f32x4 b3 = *reinterpret_cast<const f32x4*>(bp + 80);
            e8 o0, o1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o0[j] = (E)(ae[j] + (j < 4 ? b0[j & 3] : b1[j & 3]));
                o1[j] = (E)(ae[8 + j] + (j < 4 ? b2[j & 3] : b3[j & 3]));
            }
            if constexpr (WHICH == 0) {
                qf[0] = o0;
                qf[1] = o1;
            } else {
                *reinterpret_cast<e8*>(Ks + ltok * QA_KROW + 8 * h) = o0;
                *reinterpret_cast<e8*>(Ks + ltok * QA_KROW + 16 + 8 * h) = o1;
            }
        }
        QA_STAMP(t2);
        // loads and stores retire in issue order: all but this image's own weight loads have landed, the next
        // image's among them (the two stores of a head's attention output were issued before these loads)
        // (the LDS writes of the k and v images above have to have landed as well before the barrier lets others read)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(QA_DPW) : "memory");
        __builtin_amdgcn_s_barrier();
#if QA_STAMPS
        QA_STAMP(t3);
        st[0] += t1 - t0; st[1] += t2 - t1; st[2] += t3 - t2;
#endif
    };

    // both lanes of a query (32 apart) combine their halves: one cross-half swap instruction, no LDS round trip
    auto both_max = [](float v) {
        const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return fmaxf(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
    };
    auto both_sum = [](float v) {
        const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
    };
    auto attention = [&](int head) {
        if (n_mine <= 0) return;                             // uniform: a wave past the tile's tokens
        QA_STAMP(ta0);
        float m_run = QA_NEG, l_run = 0.f;
        f32x16 o = zero16;
        // S^T of one key block: rows = keys kb..kb+31 (register i is key kb + 8 (i >> 2) + 4 h + (i & 3)), column =
        // this lane's query; the code product first, into the zero accumulator
        auto scores = [&](int kb) {
            f32x16 sc = El16<E>::mfma(*reinterpret_cast<const e8*>(smem + QA_OFF_KC + ((kb + r) * 16 + 8 * h) * 2), qc, zero16);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const e8 a = *reinterpret_cast<const e8*>(Ks + (kb + r) * QA_KROW + 16 * c + 8 * h);
                sc = El16<E>::mfma(a, qf[c], sc);
            }
            return sc;
        };
        auto block_max = [&](const f32x16& sc) {
            float mx = sc[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sc[i]);
            return mx;
        };
        // weights 2^(score scale - m), their sum (this lane's half), and O^T += V^T P^T
        auto weigh = [&](f32x16& sc, int kb, float m_new) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                sc[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[i], p.scale, -m_new));
                sum += sc[i];
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                e8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[j] = (E)sc[8 * t + j];
                const E* vp = Vt + (size_t)r * QA_VROW + kb + 16 * t + 4 * h;
                const e4 lo = *reinterpret_cast<const e4*>(vp);
                const e4 hi = *reinterpret_cast<const e4*>(vp + 8);
                e8 av;
#pragma unroll
                for (int j = 0; j < 4; ++j) { av[j] = lo[j]; av[4 + j] = hi[j]; }
                o = El16<E>::mfma(av, pb, o);
            }
            return sum;
        };
        // Two key blocks per step while there are two: their score products, exponentials and value products are
        // independent of each other, and the running maximum / sum are brought up to date once per 64 keys.  Scores
        // are taken in units of 1 / scale until the exponent (max and scale commute, scale > 0).  A block may hold no
        // key of this lane's sequence: its weights are then spread over masked keys, and vanish exactly when the first
        // real key raises the running maximum by thousands (alpha = 0); a lane without any key (past the tile) is
        // never stored.
        for (int kb = kb_lo; kb < kb_hi; kb += 64) {
            const bool two = kb + 32 < kb_hi;                // uniform
            f32x16 sa2 = scores(kb), sb2 = zero16;
            float mx = block_max(sa2);
            if (two) {
                sb2 = scores(kb + 32);
                mx = fmaxf(mx, block_max(sb2));
            }
            const float m_new = fmaxf(m_run, both_max(mx) * p.scale);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // 0 at the first step: nothing to keep yet
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] *= alpha;
            float sum = weigh(sa2, kb, m_new);
            if (two) sum += weigh(sb2, kb + 32, m_new);
            l_run = l_run * alpha + both_sum(sum);
            m_run = m_new;
        }
        // O^T: rows d = 8g + 4h + j (registers), column = this lane's query; the two lanes of a query swap half of
        // their groups so that each stores 16-byte pieces (see attention_kernel)
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        uint2 grp[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            e4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (E)(o[4 * g + j] * inv);
            grp[g] = __builtin_bit_cast(uint2, v);
        }
        // (one swap instruction per register: the lower lane then holds its own group and the upper lane's, and
        // the other way round)
        auto trade = [](uint32_t& a, uint32_t& b) {
            const auto r2 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
            a = r2[0];
            b = r2[1];
        };
        trade(grp[0].x, grp[1].x); trade(grp[0].y, grp[1].y);
        trade(grp[2].x, grp[3].x); trade(grp[2].y, grp[3].y);
        const uint2 a0 = grp[0], a1 = grp[1], b0 = grp[2], b1 = grp[3];
        if (stores) {
            E* dst = p.out + (FIRST ? (int64_t)(sa + sid - 1) : T0 + ltok) * ML_HID + QA_D * head + 8 * h;
            *reinterpret_cast<uint4*>(dst) = make_uint4(a0.x, a0.y, a1.x, a1.y);
            *reinterpret_cast<uint4*>(dst + 16) = make_uint4(b0.x, b0.y, b1.x, b1.y);
        }
#if QA_STAMPS
        QA_STAMP(ta1);
        st[3] += ta1 - ta0;
#endif
    };

    for (int head = 0; head < ML_HID / QA_D; ++head) {
        image(head, std::integral_constant<int, 0>{});
        image(head, std::integral_constant<int, 1>{});
        image(head, std::integral_constant<int, 2>{});
        attention(head);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the repeats of the last image: nothing may land in LDS behind the workgroup
#if QA_STAMPS
    QA_STAMP(tk1);
    if (blockIdx.x == 300 && tid == 64) { qa_dbg[0] = st[0]; qa_dbg[1] = st[1]; qa_dbg[2] = st[2]; qa_dbg[3] = st[3]; qa_dbg[4] = tp - tk0; qa_dbg[5] = tk1 - tk0; qa_dbg[6] = (unsigned long long)(kb_hi - kb_lo); qa_dbg[7] = (unsigned long long)ntok; }
#endif
}

}  // namespace

#if QA_STAMPS
extern "C" int rag_enc_qkv_attention_debug(unsigned long long* out8) {
    RAG_HIP_TRY(hipDeviceSynchronize());
    RAG_HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(qa_dbg), 8 * sizeof(unsigned long long)));
    return RAG_OK;
}
#endif
#if LN_STAMPS
extern "C" int rag_enc_linear_debug(unsigned long long* out8) {
    RAG_HIP_TRY(hipDeviceSynchronize());
    RAG_HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(ln_dbg), 8 * sizeof(unsigned long long)));
    return RAG_OK;
}
#endif
#if ML_STAMPS
extern "C" int rag_enc_mlp_debug(unsigned long long* out8) {
    RAG_HIP_TRY(hipDeviceSynchronize());
    RAG_HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(ml_dbg), 8 * sizeof(unsigned long long)));
    return RAG_OK;
}
#endif

namespace {

// every matrix and vector is read or written in 16-byte pieces
bool aligned16(std::initializer_list<const void*> ptrs) {
    for (const void* q : ptrs)
        if (reinterpret_cast<uintptr_t>(q) % 16 != 0) return false;
    return true;
}

// Measurement aid: while enabled, every launch is bracketed by a HIP event pair on its stream.
// One state per process (one process drives one GPU here); not thread safe.
struct MlpProfile {
    static constexpr int RING = 1024;
    bool on = false;
    int n = 0;
    hipEvent_t ev0[RING], ev1[RING];
    int64_t rows[RING];
};
MlpProfile g_prof;

template <typename E>
int launch_mlp(const MlpParamsT<E>& p, int mode, hipStream_t stream, const char* who) {
    RAG_REQUIRE(ceil_div64(p.M, ML_BM) < (1ll << 31), RAG_ELIMIT, "%s: too many rows", who);
    const size_t lds_bytes = (size_t)ML_OFF_B1 + 4 * (size_t)p.I + 6 * ML_HID * 2;
    const int max_lds = ML_OFF_B1 + 4 * ML_MAX_I + 6 * ML_HID * 2;
    if (mode == 1) RAG_GRANT_LDS((mlp_fused_kernel<E, 1>), max_lds);
    else if (mode == 2) RAG_GRANT_LDS((mlp_fused_kernel<E, 2>), max_lds);
    else RAG_GRANT_LDS((mlp_fused_kernel<E, 0>), max_lds);
    const dim3 grid((unsigned)ceil_div64(p.M, ML_BM));
    const bool timed = g_prof.on && g_prof.n < MlpProfile::RING && !rag_stream_capturing(stream);
    if (timed) RAG_HIP_TRY(hipEventRecord(g_prof.ev0[g_prof.n], stream));
    if (mode == 1)
        hipLaunchKernelGGL((mlp_fused_kernel<E, 1>), grid, dim3(ML_NT), lds_bytes, stream, p);
    else if (mode == 2)
        hipLaunchKernelGGL((mlp_fused_kernel<E, 2>), grid, dim3(ML_NT), lds_bytes, stream, p);
    else
        hipLaunchKernelGGL((mlp_fused_kernel<E, 0>), grid, dim3(ML_NT), lds_bytes, stream, p);
    RAG_HIP_TRY(hipGetLastError());
    if (timed) {
        RAG_HIP_TRY(hipEventRecord(g_prof.ev1[g_prof.n], stream));
        g_prof.rows[g_prof.n++] = p.M;
    }
    return RAG_OK;
}

// the parameter block of the feed-forward kernels from the untyped pointers of the C ABI
template <typename E>
MlpParamsT<E> mlp_params(const void* x, const void* w1, const void* b1, const void* w2, const void* b2, const void* gamma,
                         const void* beta, void* y, int64_t m, int intermediate, float eps) {
    MlpParamsT<E> p = {};
    p.x = (const E*)x; p.w1 = (const E*)w1; p.b1 = (const E*)b1; p.w2 = (const E*)w2;
    p.b2 = (const E*)b2; p.gamma = (const E*)gamma; p.beta = (const E*)beta; p.y = (E*)y;
    p.M = m; p.I = intermediate; p.eps = eps;
    return p;
}

}  // namespace

extern "C" int rag_enc_mlp_dev(const void* d_x, const void* d_w1, const void* d_b1, const void* d_w2, const void* d_b2,
                               const void* d_gamma, const void* d_beta, float eps, int64_t m, int32_t hidden,
                               int32_t intermediate, int32_t dtype, void* d_y, void* stream_) {
    RAG_REQUIRE(dtype == RAG_DTYPE_BF16 || dtype == RAG_DTYPE_F16, RAG_ELIMIT, "rag_enc_mlp: bf16 or fp16 only");
    RAG_REQUIRE(hidden == ML_HID, RAG_ELIMIT, "rag_enc_mlp: hidden size must be %d (got %d)", ML_HID, hidden);
    RAG_REQUIRE(intermediate >= ML_C && intermediate % ML_C == 0 && intermediate <= ML_MAX_I, RAG_ELIMIT,
                "rag_enc_mlp: intermediate size must be a multiple of %d up to %d (got %d)", ML_C, ML_MAX_I, intermediate);
    RAG_REQUIRE(m >= 0, RAG_EINVAL, "rag_enc_mlp: negative row count");
    if (m == 0) return RAG_OK;
    RAG_REQUIRE(d_x && d_w1 && d_b1 && d_w2 && d_b2 && d_gamma && d_beta && d_y, RAG_EINVAL, "rag_enc_mlp: null pointer");
    RAG_REQUIRE(d_x != d_y, RAG_EINVAL, "rag_enc_mlp: y must not alias x (x is read again as the residual)");
    RAG_REQUIRE(aligned16({d_x, d_w1, d_w2, d_b2, d_gamma, d_beta, d_y}), RAG_EINVAL, "rag_enc_mlp: pointers must be 16-byte aligned");
    if (dtype == RAG_DTYPE_F16)
        return launch_mlp(mlp_params<f16>(d_x, d_w1, d_b1, d_w2, d_b2, d_gamma, d_beta, d_y, m, intermediate, eps), 0,
                          (hipStream_t)stream_, "rag_enc_mlp");
    return launch_mlp(mlp_params<bf16>(d_x, d_w1, d_b1, d_w2, d_b2, d_gamma, d_beta, d_y, m, intermediate, eps), 0,
                      (hipStream_t)stream_, "rag_enc_mlp");
}

extern "C" int rag_enc_attn_out_mlp_dev(const void* d_ctx, const void* d_x, const void* d_wo, const void* d_bo,
                                        const void* d_gamma1, const void* d_beta1, const void* d_w1, const void* d_b1,
                                        const void* d_w2, const void* d_b2, const void* d_gamma2, const void* d_beta2,
                                        float eps, int64_t m, int32_t hidden, int32_t intermediate, int32_t dtype,
                                        void* d_y, void* stream_) {
    RAG_REQUIRE(dtype == RAG_DTYPE_BF16 || dtype == RAG_DTYPE_F16, RAG_ELIMIT, "rag_enc_attn_out_mlp: bf16 or fp16 only");
    RAG_REQUIRE(hidden == ML_HID, RAG_ELIMIT, "rag_enc_attn_out_mlp: hidden size must be %d (got %d)", ML_HID, hidden);
    RAG_REQUIRE(intermediate >= ML_C && intermediate % ML_C == 0 && intermediate <= ML_MAX_I, RAG_ELIMIT,
                "rag_enc_attn_out_mlp: intermediate size must be a multiple of %d up to %d (got %d)", ML_C, ML_MAX_I,
                intermediate);
    RAG_REQUIRE(m >= 0, RAG_EINVAL, "rag_enc_attn_out_mlp: negative row count");
    if (m == 0) return RAG_OK;
    RAG_REQUIRE(d_ctx && d_x && d_wo && d_bo && d_gamma1 && d_beta1 && d_w1 && d_b1 && d_w2 && d_b2 && d_gamma2 && d_beta2 && d_y,
                RAG_EINVAL, "rag_enc_attn_out_mlp: null pointer");
    RAG_REQUIRE(aligned16({d_ctx, d_x, d_wo, d_bo, d_gamma1, d_beta1, d_w1, d_w2, d_b2, d_gamma2, d_beta2, d_y}), RAG_EINVAL,
                "rag_enc_attn_out_mlp: pointers must be 16-byte aligned");
    auto run = [&](auto elem) -> int {
        using E = decltype(elem);
        MlpParamsT<E> p = mlp_params<E>(d_x, d_w1, d_b1, d_w2, d_b2, d_gamma2, d_beta2, d_y, m, intermediate, eps);
        p.ctx = (const E*)d_ctx; p.wo = (const E*)d_wo; p.bo = (const E*)d_bo;
        p.gamma1 = (const E*)d_gamma1; p.beta1 = (const E*)d_beta1;
        return launch_mlp(p, 1, (hipStream_t)stream_, "rag_enc_attn_out_mlp");
    };
    return dtype == RAG_DTYPE_F16 ? run(f16{}) : run(bf16{});
}

extern "C" int rag_enc_norm_mlp_dev(const void* d_proj, const void* d_x, const void* d_gamma1, const void* d_beta1,
                                    const void* d_w1, const void* d_b1, const void* d_w2, const void* d_b2,
                                    const void* d_gamma2, const void* d_beta2, float eps, int64_t m, int32_t hidden,
                                    int32_t intermediate, int32_t dtype, void* d_y, void* stream_) {
    RAG_REQUIRE(dtype == RAG_DTYPE_BF16 || dtype == RAG_DTYPE_F16, RAG_ELIMIT, "rag_enc_norm_mlp: bf16 or fp16 only");
    RAG_REQUIRE(hidden == ML_HID, RAG_ELIMIT, "rag_enc_norm_mlp: hidden size must be %d (got %d)", ML_HID, hidden);
    RAG_REQUIRE(intermediate >= ML_C && intermediate % ML_C == 0 && intermediate <= ML_MAX_I, RAG_ELIMIT,
                "rag_enc_norm_mlp: intermediate size must be a multiple of %d up to %d (got %d)", ML_C, ML_MAX_I,
                intermediate);
    RAG_REQUIRE(m >= 0, RAG_EINVAL, "rag_enc_norm_mlp: negative row count");
    if (m == 0) return RAG_OK;
    RAG_REQUIRE(d_proj && d_x && d_gamma1 && d_beta1 && d_w1 && d_b1 && d_w2 && d_b2 && d_gamma2 && d_beta2 && d_y,
                RAG_EINVAL, "rag_enc_norm_mlp: null pointer");
    RAG_REQUIRE(aligned16({d_proj, d_x, d_gamma1, d_beta1, d_w1, d_w2, d_b2, d_gamma2, d_beta2, d_y}), RAG_EINVAL,
                "rag_enc_norm_mlp: pointers must be 16-byte aligned");
    auto run = [&](auto elem) -> int {
        using E = decltype(elem);
        MlpParamsT<E> p = mlp_params<E>(d_x, d_w1, d_b1, d_w2, d_b2, d_gamma2, d_beta2, d_y, m, intermediate, eps);
        p.ctx = (const E*)d_proj; p.gamma1 = (const E*)d_gamma1; p.beta1 = (const E*)d_beta1;
        return launch_mlp(p, 2, (hipStream_t)stream_, "rag_enc_norm_mlp");
    };
    return dtype == RAG_DTYPE_F16 ? run(f16{}) : run(bf16{});
}

extern "C" int rag_enc_mlp_profile(int32_t enable) {
    if (enable && !g_prof.on) {
        for (int i = 0; i < MlpProfile::RING; ++i) {
            RAG_HIP_TRY(hipEventCreate(&g_prof.ev0[i]));
            RAG_HIP_TRY(hipEventCreate(&g_prof.ev1[i]));
        }
        g_prof.on = true;
    }
    if (!enable && g_prof.on) {
        for (int i = 0; i < MlpProfile::RING; ++i) {
            (void)hipEventDestroy(g_prof.ev0[i]);
            (void)hipEventDestroy(g_prof.ev1[i]);
        }
        g_prof.on = false;
    }
    g_prof.n = 0;
    return RAG_OK;
}

extern "C" int rag_enc_mlp_profile_read(float* out_ms, int64_t* out_rows, int32_t max_n, int32_t* out_n) {
    RAG_REQUIRE(out_ms && out_rows && out_n, RAG_EINVAL, "rag_enc_mlp_profile_read: null argument");
    int n = g_prof.on ? g_prof.n : 0;
    if (n > max_n) n = max_n;
    for (int i = 0; i < n; ++i) {
        RAG_HIP_TRY(hipEventSynchronize(g_prof.ev1[i]));
        RAG_HIP_TRY(hipEventElapsedTime(&out_ms[i], g_prof.ev0[i], g_prof.ev1[i]));
        out_rows[i] = g_prof.rows[i];
    }
    *out_n = n;
    g_prof.n = 0;
    return RAG_OK;
}

extern "C" int rag_enc_linear384_dev(const void* d_x, const void* d_w, const void* d_bias, int64_t m, int32_t n,
                                     const void* d_residual, const void* d_gamma, const void* d_beta, float eps,
                                     int32_t dtype, void* d_y, void* stream_) {
    RAG_REQUIRE(dtype == RAG_DTYPE_BF16 || dtype == RAG_DTYPE_F16, RAG_ELIMIT, "rag_enc_linear384: bf16 or fp16 only");
    RAG_REQUIRE(n >= ML_C && n % ML_C == 0 && n <= 32768, RAG_ELIMIT,
                "rag_enc_linear384: the output width must be a multiple of %d (got %d)", ML_C, n);
    RAG_REQUIRE(m >= 0, RAG_EINVAL, "rag_enc_linear384: negative row count");
    if (m == 0) return RAG_OK;
    RAG_REQUIRE(d_x && d_w && d_bias && d_y, RAG_EINVAL, "rag_enc_linear384: null pointer");
    RAG_REQUIRE(d_x != d_y, RAG_EINVAL, "rag_enc_linear384: y must not alias x");
    RAG_REQUIRE(aligned16({d_x, d_w, d_residual, d_gamma, d_beta, d_y}), RAG_EINVAL,
                "rag_enc_linear384: pointers must be 16-byte aligned");
    RAG_REQUIRE(d_residual == nullptr || (n == ML_HID && d_gamma && d_beta), RAG_EINVAL,
                "rag_enc_linear384: the residual + LayerNorm tail needs n == %d, gamma and beta", ML_HID);
    RAG_REQUIRE(ceil_div64(m, ML_BM) < (1ll << 31), RAG_ELIMIT, "rag_enc_linear384: too many rows");
    const size_t lds_bytes = 2 * (size_t)ML_W_BYTES + 4 * (size_t)n;
    RAG_REQUIRE(lds_bytes <= 80 * 1024, RAG_ELIMIT, "rag_enc_linear384: output width %d too large", n);
    const dim3 grid((unsigned)ceil_div64(m, ML_BM));
    auto launch = [&](auto elem) -> int {
        using E = decltype(elem);
        LinearParamsT<E> p = {};
        p.x = (const E*)d_x; p.w = (const E*)d_w; p.bias = (const E*)d_bias; p.y = (E*)d_y; p.M = m; p.N = n;
        p.residual = (const E*)d_residual; p.gamma = (const E*)d_gamma; p.beta = (const E*)d_beta; p.eps = eps;
        if (d_residual) RAG_GRANT_LDS((linear384_kernel<E, true>), 80 * 1024);
        else RAG_GRANT_LDS((linear384_kernel<E, false>), 80 * 1024);
        if (d_residual)
            hipLaunchKernelGGL((linear384_kernel<E, true>), grid, dim3(ML_NT), lds_bytes, (hipStream_t)stream_, p);
        else
            hipLaunchKernelGGL((linear384_kernel<E, false>), grid, dim3(ML_NT), lds_bytes, (hipStream_t)stream_, p);
        return RAG_OK;
    };
    const int rc = dtype == RAG_DTYPE_F16 ? launch(f16{}) : launch(bf16{});
    if (rc) return rc;
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

// Cuts a packed batch into the tiles of rag_enc_qkv_attention_dev on the host: tile_seq[t] = first sequence of tile t,
// whole sequences while they sum to at most 256 tokens (greedy, in order).  h_cu and h_tile_seq are HOST arrays
// (n_seq + 1 entries each at most); returns the number of tiles in *n_tiles.
extern "C" int rag_enc_qkv_attention_tiles(const int64_t* h_cu, int32_t n_seq, int32_t* h_tile_seq, int32_t* n_tiles) {
    RAG_REQUIRE(h_cu && h_tile_seq && n_tiles && n_seq >= 0, RAG_EINVAL, "rag_enc_qkv_attention_tiles: bad arguments");
    int t = 0;
    int64_t start = n_seq ? h_cu[0] : 0;
    if (n_seq) h_tile_seq[0] = 0;
    for (int s = 0; s < n_seq; ++s) {
        const int64_t len = h_cu[s + 1] - h_cu[s];
        RAG_REQUIRE(len >= 1 && len <= QA_TILE, RAG_ELIMIT,
                    "rag_enc_qkv_attention_tiles: sequence %d has %lld tokens (1..%d supported)", s, (long long)len, QA_TILE);
        if (h_cu[s + 1] - start > QA_TILE) {
            h_tile_seq[++t] = s;
            start = h_cu[s];
        }
    }
    if (n_seq) h_tile_seq[++t] = n_seq;
    *n_tiles = t;
    return RAG_OK;
}

// Orders the sequences of a batch (h_len[i] tokens each, HOST array) so that consecutive ones fill tiles of 256 tokens
// well: a tile takes the longest remaining sequence, then again and again the longest one that still fits (lengths
// bucketed, so the whole pass is linear).  h_order[j] = index of the sequence that comes j-th, h_tile_seq[t] = position
// in that order of tile t's first sequence (n_tiles + 1 entries); sequences of no tokens come last, in no tile.
// RAG_ELIMIT if a sequence has more than 256 tokens.
extern "C" int rag_enc_qkv_attention_pack(const int32_t* h_len, int32_t n, int64_t* h_order, int32_t* h_tile_seq,
                                          int32_t* n_tiles) {
    RAG_REQUIRE(h_len && h_order && h_tile_seq && n_tiles && n >= 0, RAG_EINVAL, "rag_enc_qkv_attention_pack: bad arguments");
    int32_t count[QA_TILE + 2] = {0};
    for (int i = 0; i < n; ++i) {
        RAG_REQUIRE(h_len[i] >= 0 && h_len[i] <= QA_TILE, RAG_ELIMIT,
                    "rag_enc_qkv_attention_pack: sequence %d has %d tokens (at most %d supported)", i, h_len[i], QA_TILE);
        ++count[h_len[i] + 1];
    }
    for (int l = 0; l <= QA_TILE; ++l) count[l + 1] += count[l];          // count[l] = first slot of length l
    std::vector<int32_t> by_len((size_t)n);
    {
        std::vector<int32_t> at(count, count + QA_TILE + 1);
        for (int i = 0; i < n; ++i) by_len[(size_t)at[h_len[i]]++] = i;    // stable: input order inside a length
    }
    int32_t cur[QA_TILE + 1];                                             // next unused slot of every length
    uint64_t have[(QA_TILE + 64) / 64] = {0};                             // bit l: length l has unused sequences
    for (int l = 1; l <= QA_TILE; ++l) {
        cur[l] = count[l];
        if (count[l + 1] > count[l]) have[l >> 6] |= 1ull << (l & 63);
    }
    auto longest_up_to = [&](int cap) -> int {                            // 0 if none
        for (int w = cap >> 6; w >= 0; --w) {
            uint64_t m = have[w];
            if (w == (cap >> 6)) m &= (cap & 63) == 63 ? ~0ull : ((1ull << ((cap & 63) + 1)) - 1);
            if (m) return (w << 6) + 63 - __builtin_clzll(m);
        }
        return 0;
    };
    int64_t pos = 0;
    int t = 0;
    for (;;) {
        int l = longest_up_to(QA_TILE);
        if (l == 0) break;
        h_tile_seq[t++] = (int32_t)pos;
        int cap = QA_TILE;
        while (l > 0) {
            h_order[pos++] = by_len[(size_t)cur[l]++];
            if (cur[l] == count[l + 1]) have[l >> 6] &= ~(1ull << (l & 63));
            cap -= l;
            l = cap > 0 ? longest_up_to(cap) : 0;
        }
    }
    h_tile_seq[t] = (int32_t)pos;
    for (int i = count[0]; i < count[1]; ++i) h_order[pos++] = by_len[(size_t)i];
    *n_tiles = t;
    return RAG_OK;
}

namespace {
int qkv_attention_launch(const void* d_x, const void* d_w, const void* d_bias, const int32_t* d_cu_seqlens, int32_t n_seq,
                         int64_t total_tokens, const int32_t* d_tile_seq, int32_t n_tiles, int32_t n_heads, int32_t head_dim,
                         int32_t dtype, void* d_out, void* stream_, bool first_only) {
    RAG_REQUIRE(dtype == RAG_DTYPE_BF16 || dtype == RAG_DTYPE_F16, RAG_ELIMIT, "rag_enc_qkv_attention: bf16 or fp16 only");
    RAG_REQUIRE(head_dim == QA_D && n_heads * head_dim == ML_HID, RAG_ELIMIT,
                "rag_enc_qkv_attention: %d heads of %d (got %d of %d)", ML_HID / QA_D, QA_D, n_heads, head_dim);
    RAG_REQUIRE(n_seq >= 0 && n_tiles >= 0 && n_tiles <= n_seq, RAG_EINVAL, "rag_enc_qkv_attention: bad counts");
    RAG_REQUIRE(total_tokens >= 0 && total_tokens < (1ll << 31), RAG_ELIMIT, "rag_enc_qkv_attention: bad token count");
    if (n_seq == 0 || n_tiles == 0) return RAG_OK;
    RAG_REQUIRE(d_x && d_w && d_bias && d_cu_seqlens && d_tile_seq && d_out, RAG_EINVAL, "rag_enc_qkv_attention: null pointer");
    RAG_REQUIRE(d_x != d_out, RAG_EINVAL, "rag_enc_qkv_attention: out must not alias x");
    RAG_REQUIRE(aligned16({d_x, d_w, d_out}), RAG_EINVAL, "rag_enc_qkv_attention: pointers must be 16-byte aligned");
    uint32_t* fault = rag_fault_word();
    RAG_REQUIRE(fault != nullptr, RAG_EINVAL, "rag_enc_qkv_attention: call rag_init(device) first");
    auto launch = [&](auto elem) {
        using E = decltype(elem);
        QkvAttnParams<E> p = {};
        p.x = (const E*)d_x; p.w = (const E*)d_w; p.bias = (const E*)d_bias; p.out = (E*)d_out;
        p.cu = d_cu_seqlens; p.tile_seq = d_tile_seq; p.T = total_tokens; p.n_seq = n_seq;
        p.scale = 1.4426950408889634f / sqrtf((float)head_dim); p.fault = fault;
        if (first_only) {
            RAG_GRANT_LDS((qkv_attention_kernel<E, true>), QA_LDS);
            hipLaunchKernelGGL((qkv_attention_kernel<E, true>), dim3((unsigned)n_tiles), dim3(QA_NT), QA_LDS, (hipStream_t)stream_, p);
        } else {
            RAG_GRANT_LDS((qkv_attention_kernel<E, false>), QA_LDS);
            hipLaunchKernelGGL((qkv_attention_kernel<E, false>), dim3((unsigned)n_tiles), dim3(QA_NT), QA_LDS, (hipStream_t)stream_, p);
        }
        return RAG_OK;
    };
    const int rc = dtype == RAG_DTYPE_F16 ? launch(f16{}) : launch(bf16{});
    if (rc) return rc;
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}
}  // namespace

extern "C" int rag_enc_qkv_attention_dev(const void* d_x, const void* d_w, const void* d_bias, const int32_t* d_cu_seqlens,
                                         int32_t n_seq, int64_t total_tokens, const int32_t* d_tile_seq, int32_t n_tiles,
                                         int32_t n_heads, int32_t head_dim, int32_t dtype, void* d_out, void* stream_) {
    return qkv_attention_launch(d_x, d_w, d_bias, d_cu_seqlens, n_seq, total_tokens, d_tile_seq, n_tiles, n_heads, head_dim,
                                dtype, d_out, stream_, false);
}

extern "C" int rag_enc_qkv_attention_first_dev(const void* d_x, const void* d_w, const void* d_bias,
                                               const int32_t* d_cu_seqlens, int32_t n_seq, int64_t total_tokens,
                                               const int32_t* d_tile_seq, int32_t n_tiles, int32_t n_heads, int32_t head_dim,
                                               int32_t dtype, void* d_out, void* stream_) {
    return qkv_attention_launch(d_x, d_w, d_bias, d_cu_seqlens, n_seq, total_tokens, d_tile_seq, n_tiles, n_heads, head_dim,
                                dtype, d_out, stream_, true);
}
