// Second half of a 384-wide BERT layer's feed-forward block — and, with another operand source, the
// attention-output projection — at the reference's float32 precision on the 16-bit matrix cores:
//
//   Y = LayerNorm( R + act(H + b1) W^T + b2 ) * gamma + beta          W [384][K], fp32 in, fp32 out
//
// Every operand of the product is held as two fp16 halves, x = hi + lo (22 significant bits), and every
// product is taken as hi.hi + lo.hi + hi.lo with fp32 accumulation (v_mfma_f32_32x32x16_f16): the result is
// as good as an fp32 product (see rag_enc_split_bf16x3_dev in rag_hip.h for the measurement).  Replaces, for
// the float32 cross-encoder forward (reference: CrossEncoder.predict, src/reranker/rerank.py:79-84), a library
// GEMM over a [tokens][3K] operand plus two memory-bound passes (bias + GELU + split in front of it,
// bias + residual + LayerNorm + split behind it): per token the block now reads 4K + 1536 bytes and writes
// 1536 + 2304 instead of moving 4K + 6K + 6K + 3072 + ... through HBM three times.
//
// Structure (the second product of mlp_fused_kernel, mlp.hip, on its own): one workgroup of four waves, one
// per SIMD, owns 128 tokens; a wave owns 32 of them.  The product is computed transposed,
// Y^T[feature][token] += W[feature][k] . A^T[k][token]: the weights are the MFMA's A operand out of LDS (hi and
// lo images of 32 k-columns x 384 features, double buffered, filled by direct-to-LDS loads, XOR-swizzled),
// the activations its B operand out of registers: a lane holds its token's 16 values of a k-chunk, which it
// loaded from global memory one chunk ahead and turned into halves (bias, GELU, split) between the MFMAs of
// the chunk before.  Twelve 32x32 accumulator tiles (192 registers) hold Y^T; the feature order of the
// weight images is chosen so that a lane's accumulator registers are runs of eight consecutive features: bias,
// residual, gamma, beta and the output are 32-byte accesses, LayerNorm statistics need one half-wave shuffle.
#include "common.h"

#include <math.h>
#include <initializer_list>
#include <type_traits>
#include <utility>

namespace {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int FS_HID = 384;                  // output width (the layer's hidden size)
constexpr int FS_BM = 128;                   // tokens per workgroup
constexpr int FS_NT = 256;                   // four waves
constexpr int FS_C = 32;                     // k-columns per chunk
constexpr int FS_YT = FS_HID / 32;           // accumulator tiles: 12
constexpr int FS_STEPS = 2 * FS_YT;          // (k-step, tile) pairs per chunk: 24
constexpr int FS_W_BYTES = FS_C * FS_HID * 2;              // one image (hi or lo) of a chunk: 24576 B
constexpr int FS_W_DMAS = FS_W_BYTES / 1024 / (FS_NT / 64);   // 1-KiB direct-to-LDS loads per wave and image: 6

constexpr int FS_T_ROW = FS_HID + 4;          // floats per row of the epilogue's transposition buffer (1552 bytes)
constexpr int FS_MAX_K = 2048;               // b1 sits in LDS beside the buffers
constexpr int FS_OFF_OP = 4 * FS_W_BYTES;    // behind the two (hi, lo) weight buffers: the operand ring
constexpr int FS_OP_WAVE = 4096;             // a wave's 32 tokens x 32 k-columns: fp32, or fp16 high + low
constexpr int FS_OP_SLOT = (FS_NT / 64) * FS_OP_WAVE;
constexpr int FS_OFF_B1 = FS_OFF_OP + 3 * FS_OP_SLOT;   // behind the ring: b1 [K] fp32, then b2 | gamma | beta [384] fp32

#ifndef FS_STAMPS
#define FS_STAMPS 0
#endif
#if FS_STAMPS
__device__ unsigned long long fs_dbg[8];
#define FS_STAMP(var) unsigned long long var = __builtin_readcyclecounter()
#else
#define FS_STAMP(var)
#endif

struct FfnOutParams {
    const float* h;        // SRC 0: [M][K] fp32, the first product without its bias
    const f16* hs;         // SRC 1: fp16, a finished operand already split: [M][3K] = [hi | hi | lo], or [M][2K] = [hi | lo]
    int hs_hl;             // SRC 1: rows are [hi | lo]
    const float* b1;       // SRC 0: [K] bias added before the activation
    const f16* wh;         // high halves of W [384][K], packed by chunk (pack_weights_kernel)
    const f16* wl;         // low halves, packed the same way
    const float* b2;       // [384]
    const float* residual; // [M][384], or null with
    const f16* residual_s; // [M][1152] fp16 rows [hi | hi | lo]: the residual is hi + lo (the ys of the call that made it)
    const float* gamma;    // [384]
    const float* beta;     // [384]
    float* y;              // [M][384], or null (only ys is wanted)
    f16* ys;               // [M][1152] = [hi | hi | lo] of y, or null
    int64_t M;
    int K;
    float eps;
};

// One wave-instruction of a direct-to-LDS load: lane l's 16 bytes at gbase + voff land at LDS byte
// lds_dst + 16 l.  M0 carries the destination and belongs to the compiler: saved and restored.  The compiler
// does not count this load: wait with s_waitcnt vmcnt before reading.
__device__ __forceinline__ void lds_dma16(uint32_t lds_dst, const void* gbase, uint32_t voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(gbase) : "memory");
}
// LDS reads whose completion the source waits for (counted lgkmcnt), so they can run several MFMAs ahead.
template <int OFF, typename T>
__device__ __forceinline__ void lds_read16(T& dst, uint32_t addr) {
    static_assert(sizeof(T) == 16 && OFF >= 0 && OFF < 65536, "16-byte read, 16-bit offset");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N, typename T>
__device__ __forceinline__ void lds_wait(T& a, T& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int... Is, typename F>
__device__ __forceinline__ void unrolled(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}

// GELU(v) = v (1 + erf(v / sqrt 2)) / 2 with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7; against
// a float64 GELU: 4.6e-7 at most, 5.9e-8 on average over [-9, 9], which is what an fp32 erf-based GELU gives
// too) — no branches, one reciprocal and one exponential.
__device__ __forceinline__ float gelu_f32(float v) {
    const float z = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float q = fmaf(1.061405429f, t, -1.453152027f);
    q = fmaf(q, t, 1.421413741f);
    q = fmaf(q, t, -0.284496736f);
    q = fmaf(q, t, 0.254829592f);
    q *= t;
    const float e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
    const float erf_abs = fmaf(-q, e, 1.0f);
    const float erf = copysignf(erf_abs, v);
    return 0.5f * v * (1.0f + erf);
}

// x -> fp16 halves; the opaque register keeps the compiler from folding the arithmetic that produced x into
// one of the two conversions (v_fma_mixlo_f16), which would make the stored hi differ from the subtracted one
__device__ __forceinline__ void split_f16(float x, f16& hi, f16& lo) {
    asm volatile("" : "+v"(x));
    hi = (f16)x;
    lo = (f16)(x - (float)hi);
}

// The same GELU on two values at once: the polynomial runs on the packed fp32 instructions (v_pk_fma_f32, ...),
// half the issue slots of the scalar form; every operation is the scalar form's, so are the results.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 gelu_f32x2(f32x2 v) {
    const f32x2 z = __builtin_elementwise_abs(v) * 0.70710678118654752f;
    const f32x2 d = fma2((f32x2)(0.3275911f), z, (f32x2)(1.0f));
    const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    f32x2 q = fma2((f32x2)(1.061405429f), t, (f32x2)(-1.453152027f));
    q = fma2(q, t, (f32x2)(1.421413741f));
    q = fma2(q, t, (f32x2)(-0.284496736f));
    q = fma2(q, t, (f32x2)(0.254829592f));
    q = q * t;
    const f32x2 zz = z * z * -1.4426950408889634f;
    const f32x2 e = {__builtin_amdgcn_exp2f(zz.x), __builtin_amdgcn_exp2f(zz.y)};
    const f32x2 erf_abs = fma2(-q, e, (f32x2)(1.0f));
    const f32x2 erf = {copysignf(erf_abs.x, v.x), copysignf(erf_abs.y, v.y)};
    return 0.5f * v * (1.0f + erf);
}

// SRC 0: B operand = halves of GELU(h + b1), h fp32 from memory.  SRC 1: B operand = halves read from memory.
template <int SRC>
__global__ __launch_bounds__(FS_NT, 1) void ffn_out_split_kernel(FfnOutParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    FS_STAMP(t_start);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int64_t tok_wg = (int64_t)blockIdx.x * FS_BM;                 // first token of the workgroup
    const int rows_live = (int)min((int64_t)FS_BM, p.M - tok_wg);       // >= 1; rows past M load the last live row
    const int n_chunks = p.K / FS_C;
    const int last = n_chunks - 1;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

    // ---- weight staging.  The weights arrive packed (pack_weights_kernel below): chunk c of an image is 24 KiB of
    // contiguous memory already in LDS order, so a direct-to-LDS load moves 1 KiB of whole lines.  Image row f
    // holds the chunk's 32 k-columns of feature (f with bits 2 and 3 swapped), piece j (16 bytes = 8 columns) at
    // j ^ ((f >> 2) & 3): with that row order registers 8t..8t+7 of accumulator tile m are features
    // 32m + 16t + 8h + 0..7 of the lane's token.
    auto dma_w = [&](int c, int buf, int i, int lo) {         // piece group i of chunk c, one image -> buffer buf
        lds_dma16(lds0 + (buf * 2 + lo) * FS_W_BYTES + 1024 * (wave + 4 * i),
                  reinterpret_cast<const unsigned char*>(lo ? p.wl : p.wh) + (size_t)c * FS_W_BYTES + 1024 * (wave + 4 * i),
                  (uint32_t)(16 * lane));
    };

    // ---- operand staging: a wave's 32 tokens x 32 k-columns of a chunk (4 KiB either way) come through LDS too,
    // as four direct-to-LDS loads whose lanes walk along the rows (whole 64- or 128-byte runs of a row per 4 or 8
    // lanes; a lane-per-token load would touch 64 lines per instruction and keep the address unit busy longer than
    // the chunk's MFMAs take), into a ring of three slots so that a chunk is asked for two iterations before use.
    //   SRC 0: rows of 128 bytes (32 fp32), piece j of row w at j ^ ((w >> 1) & 7); load i brings rows 8i..8i+7.
    //   SRC 1: high image, rows of 64 bytes, piece j at j ^ ((w >> 2) & 3), then the low image 2 KiB further;
    //          loads 0, 1 bring rows 0-15, 16-31 of the high image, loads 2, 3 those of the low image.
    constexpr int CHUNK_BYTES = SRC == 0 ? FS_C * 4 : FS_C * 2;
    constexpr int N_OPOFF = SRC == 0 ? 4 : 2;
    const size_t row_bytes = SRC == 0 || p.hs_hl ? (size_t)p.K * 4 : (size_t)p.K * 6;
    const size_t lo_bytes = p.hs_hl ? 2 * (size_t)p.K : 4 * (size_t)p.K;         // SRC 1: where a row's low halves start
    const unsigned char* op_base = (SRC == 0 ? reinterpret_cast<const unsigned char*>(p.h) : reinterpret_cast<const unsigned char*>(p.hs))
                                   + (size_t)tok_wg * row_bytes;
    uint32_t opoff[N_OPOFF];
#pragma unroll
    for (int i = 0; i < N_OPOFF; ++i) {
        const int w = SRC == 0 ? 8 * i + lane / 8 : 16 * i + lane / 4;
        const int j = SRC == 0 ? (lane % 8) ^ ((w >> 1) & 7) : (lane % 4) ^ ((w >> 2) & 3);
        opoff[i] = (uint32_t)(min(wave * 32 + w, rows_live - 1) * row_bytes + 16 * j);
    }
    const uint32_t op_lds = lds0 + FS_OFF_OP + wave * FS_OP_WAVE;
    auto dma_op = [&](int c, int slot, auto i_c) {
        constexpr int I = decltype(i_c)::value;
        const uint32_t dst = op_lds + slot * FS_OP_SLOT + 1024 * I;
        if constexpr (SRC == 0) lds_dma16(dst, op_base + (size_t)c * CHUNK_BYTES, opoff[I]);
        else lds_dma16(dst, op_base + (I >= 2 ? lo_bytes : 0) + (size_t)c * CHUNK_BYTES, opoff[I & 1]);
    };
    auto dma_op_all = [&](int c, int slot) {
        unrolled(std::make_integer_sequence<int, 4>{}, [&](auto i_c) { dma_op(c, slot, i_c); });
    };
    // where the lane's values of a slot sit.  SRC 0: the eight fp32 of k-step t = pieces 4t + 2h, 4t + 2h + 1 of row r
    // (the second 16 bytes beside the first: address ^ 16).  SRC 1: the halves of k-step t = piece 2t + h of row r.
    uint32_t opl[2], opx[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        opl[t] = SRC == 0 ? op_lds + r * 128 + 16 * ((4 * t + 2 * h) ^ ((r >> 1) & 7))
                          : op_lds + r * 64 + 16 * ((2 * t + h) ^ ((r >> 2) & 3));
        opx[t] = op_lds + r * 128 + 16 * ((4 * t + 2 * h + 1) ^ ((r >> 1) & 7));      // SRC 0 only
    }

    // first loads: weights of chunk 0, the operand chunks the first iteration expects in flight
    constexpr int PF = SRC == 0 ? 3 : 2;                      // iteration c asks for operand chunk c + PF
#pragma unroll
    for (int i = 0; i < FS_W_DMAS; ++i) { dma_w(0, 0, i, 0); dma_w(0, 0, i, 1); }
#pragma unroll
    for (int c0 = 0; c0 < PF; ++c0) dma_op_all(min(c0, last), c0);

    // the vectors: b1 [K] | b2 | gamma | beta behind the buffers, every load issued before the first LDS write
    float* vec = reinterpret_cast<float*>(smem + FS_OFF_B1);
    {
        f32x4 st[4];
        const int k4 = p.K / 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + FS_NT * u;
            if (SRC == 0 && i < k4) st[u] = reinterpret_cast<const f32x4*>(p.b1)[i];
            if (i < 3 * FS_HID / 4) {
                const float* src = i < FS_HID / 4 ? p.b2 : i < 2 * FS_HID / 4 ? p.gamma : p.beta;
                st[2 + u] = reinterpret_cast<const f32x4*>(src)[i % (FS_HID / 4)];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + FS_NT * u;
            if (SRC == 0 && i < k4) reinterpret_cast<f32x4*>(vec)[i] = st[u];
            if (i < 3 * FS_HID / 4) reinterpret_cast<f32x4*>(vec + p.K)[i] = st[2 + u];
        }
    }

    // fragment address of step (t, m): piece 2t + h of image row 32 m + r; one base register per image (buffer,
    // half) and k-step, the tile as the read's immediate offset (16 bits)
    uint32_t l2[4][2];
#pragma unroll
    for (int img = 0; img < 4; ++img)
#pragma unroll
        for (int t = 0; t < 2; ++t)
            l2[img][t] = lds0 + img * FS_W_BYTES + r * (FS_C * 2) + 16 * ((2 * t + h) ^ ((r >> 2) & 3));

    struct Raw { u32x4 v[4]; };
    // the lane's sixteen values of the chunk in ring slot `slot`, through counted reads: SRC 0 raw fp32 (v[2t],
    // v[2t+1] = k-step t), SRC 1 the halves (v[t] high, v[2 + t] low)
    auto read_op = [&](int slot, Raw& q) {
        const uint32_t a0 = opl[0] + slot * FS_OP_SLOT, a1 = opl[1] + slot * FS_OP_SLOT;
        if constexpr (SRC == 0) {
            lds_read16<0>(q.v[0], a0);
            lds_read16<0>(q.v[1], opx[0] + slot * FS_OP_SLOT);
            lds_read16<0>(q.v[2], a1);
            lds_read16<0>(q.v[3], opx[1] + slot * FS_OP_SLOT);
        } else {
            lds_read16<0>(q.v[0], a0);
            lds_read16<0>(q.v[1], a1);
            lds_read16<2048>(q.v[2], a0);
            lds_read16<2048>(q.v[3], a1);
        }
    };
    // values i, i + 1 (i even) of a chunk -> their halves (SRC 0): bias, GELU, split.  bias[i >> 2][i & 3] = b1 of value i.
    auto act_two = [&](const f32x4* bias, const Raw& q, int i, f16x8* ph, f16x8* pl) {
        const f32x4 qv = __builtin_bit_cast(f32x4, q.v[i >> 2]);
        const f32x2 v = {qv[i & 3] + bias[i >> 2][i & 3], qv[(i & 3) + 1] + bias[i >> 2][(i & 3) + 1]};
        f32x2 g = gelu_f32x2(v);
        asm volatile("" : "+v"(g));
        const f16 h0 = (f16)g.x, h1 = (f16)g.y;
        ph[i >> 3][i & 7] = h0;
        ph[i >> 3][(i & 7) + 1] = h1;
        pl[i >> 3][i & 7] = (f16)(g.x - (float)h0);
        pl[i >> 3][(i & 7) + 1] = (f16)(g.y - (float)h1);
    };

    f32x16 yacc[FS_YT];
#pragma unroll
    for (int m = 0; m < FS_YT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) yacc[m][i] = 0.f;

    // ---- prologue.  SRC 0: the halves of chunk 0 are made here; the loop makes those of chunk c + 1 during chunk c.
    f16x8 phA[2], plA[2], phB[2], plB[2];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                          // weights of chunk 0, the first operand chunks and the vectors are in LDS
    const uint32_t lb = lds0 + FS_OFF_B1 + 32 * h;           // b1 of the lane's columns: chunk c at + 128 c, k-step t at + 64 t
    if constexpr (SRC == 0) {
        Raw raw0;
        read_op(0, raw0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw0.v[0]), "+v"(raw0.v[1]), "+v"(raw0.v[2]), "+v"(raw0.v[3]));
        f32x4 bias0[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) bias0[u] = *reinterpret_cast<const f32x4*>(smem + FS_OFF_B1 + 32 * h + 64 * (u >> 1) + 16 * (u & 1));
#pragma unroll
        for (int i = 0; i < 16; i += 2) act_two(bias0, raw0, i, phA, plA);
    }

    // ---- main loop.  Two weight buffers: iteration c multiplies chunk c out of buffer c % 2 while the weights of
    // chunk c + 1 stream into the other one, which every wave left at the barrier that closed iteration c - 1.
    // Operand ring: SRC 1 reads its halves of chunk c from slot c % 3 when the iteration starts; SRC 0 reads the raw
    // values of chunk c + 1 from slot (c + 1) % 3 and turns them into halves one value per block of MFMAs.  Either
    // way the slot read one iteration ago is free, and operand chunk c + PF goes there.  A wave's vector-memory
    // operations of an iteration, in order: 12 weight pieces, 4 operand pieces, one in front of each of the first
    // sixteen blocks of four MFMAs.  They complete in order, so "all but the 4 youngest" before the closing barrier
    // covers the weights of chunk c + 1 and every operand piece of the iterations before.  Past the last chunk the
    // loads repeat it, into buffers nobody reads any more.
    constexpr int FS_GROUPS = FS_STEPS / 4;                   // 6 groups of four accumulator tiles
    int slot_rd = SRC == 0 ? 1 : 0, slot_wr = PF % 3;
    auto iteration = [&](int c, auto buf_c, f16x8* ph, f16x8* pl, f16x8* nh, f16x8* nl) {
        constexpr int BUF = decltype(buf_c)::value;
        const int cn = min(c + 1, last), cpf = min(c + PF, last);
        f32x4 bias[4];
        Raw raw;
        if constexpr (SRC == 0) {                             // b1 of chunk c + 1
            const uint32_t lbc = lb + (uint32_t)cn * (FS_C * 4);
            lds_read16<0>(bias[0], lbc);
            lds_read16<16>(bias[1], lbc);
            lds_read16<64>(bias[2], lbc);
            lds_read16<80>(bias[3], lbc);
        }
        read_op(slot_rd, raw);
        // fragments of a group: tiles 4g..4g+3 (mod 12) of k-step g / 3, high and low images; two groups in flight
        u32x4 qh[2][4], ql[2][4];
        auto issue = [&](auto g_c) {
            constexpr int G = decltype(g_c)::value;
            constexpr int T = G / 3, M0 = 4 * (G % 3);
            lds_read16<(M0 + 0) * 2048>(qh[G & 1][0], l2[2 * BUF][T]);
            lds_read16<(M0 + 0) * 2048>(ql[G & 1][0], l2[2 * BUF + 1][T]);
            lds_read16<(M0 + 1) * 2048>(qh[G & 1][1], l2[2 * BUF][T]);
            lds_read16<(M0 + 1) * 2048>(ql[G & 1][1], l2[2 * BUF + 1][T]);
            lds_read16<(M0 + 2) * 2048>(qh[G & 1][2], l2[2 * BUF][T]);
            lds_read16<(M0 + 2) * 2048>(ql[G & 1][2], l2[2 * BUF + 1][T]);
            lds_read16<(M0 + 3) * 2048>(qh[G & 1][3], l2[2 * BUF][T]);
            lds_read16<(M0 + 3) * 2048>(ql[G & 1][3], l2[2 * BUF + 1][T]);
        };
        issue(std::integral_constant<int, 0>{});
        // the bias and operand reads came before the first 8 fragment reads
        if constexpr (SRC == 0)
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2]), "+v"(bias[3]), "+v"(raw.v[0]),
                         "+v"(raw.v[1]), "+v"(raw.v[2]), "+v"(raw.v[3]));
        else {
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(raw.v[0]), "+v"(raw.v[1]), "+v"(raw.v[2]), "+v"(raw.v[3]));
#pragma unroll
            for (int t = 0; t < 2; ++t) { ph[t] = __builtin_bit_cast(f16x8, raw.v[t]); pl[t] = __builtin_bit_cast(f16x8, raw.v[2 + t]); }
        }
        auto group = [&](auto g_c) {
            constexpr int G = decltype(g_c)::value;
            constexpr int T = G / 3, M0 = 4 * (G % 3), R = G & 1;
            // a block: its vector-memory operation, four MFMAs of one product over the group's tiles (an accumulator
            // is written again twelve MFMAs later) and, SRC 0, one value of the next chunk turned into halves
            auto block = [&](auto s_c, const u32x4* q, const f16x8& pb) {
                constexpr int S = decltype(s_c)::value, B = 3 * G + S;
                if constexpr (B < 2 * FS_W_DMAS) dma_w(cn, BUF ^ 1, B / 2, B % 2);
                else if constexpr (B < 2 * FS_W_DMAS + 4) dma_op(cpf, slot_wr, std::integral_constant<int, B - 2 * FS_W_DMAS>{});
                if constexpr (S == 0) {
                    if constexpr (G + 1 < FS_GROUPS) issue(std::integral_constant<int, G + 1>{});
                    constexpr int newer = G + 1 < FS_GROUPS ? 8 : 0;
                    asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(qh[R][0]), "+v"(ql[R][0]), "+v"(qh[R][1]), "+v"(ql[R][1]),
                                 "+v"(qh[R][2]), "+v"(ql[R][2]), "+v"(qh[R][3]), "+v"(ql[R][3]) : "n"(newer));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    yacc[M0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, q[j]), pb, yacc[M0 + j], 0, 0, 0);
                if constexpr (SRC == 0 && B < 16 && B % 2 == 0) act_two(bias, raw, B, nh, nl);
            };
            block(std::integral_constant<int, 0>{}, ql[R], ph[T]);
            block(std::integral_constant<int, 1>{}, qh[R], pl[T]);
            block(std::integral_constant<int, 2>{}, qh[R], ph[T]);
            if constexpr (SRC == 0) asm volatile("" : "+v"(nh[0]), "+v"(nl[0]), "+v"(nh[1]), "+v"(nl[1]));
        };
        unrolled(std::make_integer_sequence<int, FS_GROUPS>{}, group);
        slot_rd = slot_rd == 2 ? 0 : slot_rd + 1;
        slot_wr = slot_wr == 2 ? 0 : slot_wr + 1;
        asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    };
    FS_STAMP(t_loop0);
    for (int c = 0; c < n_chunks; c += 2) {
        iteration(c, std::integral_constant<int, 0>{}, phA, plA, phB, plB);
        if (c + 1 < n_chunks) iteration(c + 1, std::integral_constant<int, 1>{}, phB, plB, phA, plA);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the repeated loads of the last iterations: nothing may still be
                                                              // on its way into LDS when the buffers get their next use

    FS_STAMP(t_loop1);
    // ---- epilogue: bias, residual, LayerNorm, outputs.  The accumulators hold Y^T (a lane: 192 scattered features of
    // one token); global memory wants rows.  Each wave turns its 32 tokens through its own piece of the (now free)
    // weight buffers, 16 tokens at a time: lanes of those tokens write their runs of eight features into rows of
    // FS_T_ROW floats (row stride = 16 bytes mod 256: conflict-free both ways), then eight lanes per token read a row
    // back as six 32-byte units each, unit 8 j + (lane % 8): a wave instruction then touches whole 128-byte lines of
    // eight rows of the residual and of the three outputs.  Statistics: the lane's 48 values, then three shuffles.
    __syncthreads();                                          // every wave has left the weight buffers
    float* tb = reinterpret_cast<float*>(smem) + wave * (16 * FS_T_ROW);
    const int tk = lane >> 3, l8 = lane & 7;
    const int64_t tok0 = (int64_t)blockIdx.x * FS_BM + wave * 32 + tk;     // + 8 q: the lane's token in pass q
    const float* b2s = vec + p.K + 8 * l8;
    const float* gs = b2s + FS_HID;
    const float* bs = gs + FS_HID;
    // residual of the pass ahead, as loaded: 8 floats, or (split residual) 8 high halves and 8 low halves; turned into
    // floats where it is used, so that the loads are nothing but loads and are all in flight at once
    f32x4 rs[12];
    auto load_residual = [&](int64_t tok) {
        if (p.residual_s != nullptr) {
            const f16* res = p.residual_s + min(tok, p.M - 1) * (3 * FS_HID) + 8 * l8;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                rs[2 * j] = *reinterpret_cast<const f32x4*>(res + 64 * j);
                rs[2 * j + 1] = *reinterpret_cast<const f32x4*>(res + 2 * FS_HID + 64 * j);
            }
        } else {
            const float* res = p.residual + min(tok, p.M - 1) * FS_HID + 8 * l8;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                rs[2 * j] = *reinterpret_cast<const f32x4*>(res + 64 * j);
                rs[2 * j + 1] = *reinterpret_cast<const f32x4*>(res + 64 * j + 4);
            }
        }
    };
    load_residual(tok0);
#pragma nounroll
    for (int half = 0; half < 2; ++half) {
        if ((r >> 4) == half) {
            float* row = tb + (r & 15) * FS_T_ROW + 8 * h;
#pragma unroll
            for (int m = 0; m < FS_YT; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 a, b;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { a[j] = yacc[m][8 * t + j]; b[j] = yacc[m][8 * t + 4 + j]; }
                    *reinterpret_cast<f32x4*>(row + 32 * m + 16 * t) = a;
                    *reinterpret_cast<f32x4*>(row + 32 * m + 16 * t + 4) = b;
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma nounroll
        for (int pp = 0; pp < 2; ++pp) {
            const int q = 2 * half + pp;
            const int64_t tok_q = tok0 + 8 * q;
            const float* trow = tb + (8 * pp + tk) * FS_T_ROW + 8 * l8;
            float v[48];
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const f32x4 y0 = *reinterpret_cast<const f32x4*>(trow + 64 * j);
                const f32x4 y1 = *reinterpret_cast<const f32x4*>(trow + 64 * j + 4);
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(b2s + 64 * j);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(b2s + 64 * j + 4);
                f32x4 r0 = rs[2 * j], r1 = rs[2 * j + 1];
                if (p.residual_s != nullptr) {                // hi + lo: exact in fp32
                    const f16x8 a = __builtin_bit_cast(f16x8, rs[2 * j]), b = __builtin_bit_cast(f16x8, rs[2 * j + 1]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        r0[i] = (float)a[i] + (float)b[i];
                        r1[i] = (float)a[4 + i] + (float)b[4 + i];
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[8 * j + i] = (y0[i] + c0[i]) + r0[i];
                    v[8 * j + 4 + i] = (y1[i] + c1[i]) + r1[i];
                    sum += v[8 * j + i];
                    sum += v[8 * j + 4 + i];
                }
            }
            if (q < 3) load_residual(tok_q + 8);             // the next pass's residual, in flight during this pass
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            sum += __shfl_xor(sum, 4, 64);
            const float mean = sum / (float)FS_HID;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < 48; ++i) { const float d = v[i] - mean; sq += d * d; }
            sq += __shfl_xor(sq, 1, 64);
            sq += __shfl_xor(sq, 2, 64);
            sq += __shfl_xor(sq, 4, 64);
            const float rstd = rsqrtf(sq / (float)FS_HID + p.eps);
            const bool live = tok_q < p.M;
            float* yrow = p.y + tok_q * FS_HID + 8 * l8;
            f16* srow = p.ys != nullptr ? p.ys + tok_q * 3 * FS_HID + 8 * l8 : nullptr;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(gs + 64 * j);
                const f32x4 g1 = *reinterpret_cast<const f32x4*>(gs + 64 * j + 4);
                const f32x4 e0 = *reinterpret_cast<const f32x4*>(bs + 64 * j);
                const f32x4 e1 = *reinterpret_cast<const f32x4*>(bs + 64 * j + 4);
                f32x4 o0, o1;
                f16x8 oh, ol;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float o = (v[8 * j + i] - mean) * rstd * (i < 4 ? g0[i & 3] : g1[i & 3]) + (i < 4 ? e0[i & 3] : e1[i & 3]);
                    if (i < 4) o0[i & 3] = o; else o1[i & 3] = o;
                    f16 a, b;
                    split_f16(o, a, b);
                    oh[i] = a;
                    ol[i] = b;
                }
                if (live) {
                    if (p.y != nullptr) {
                        *reinterpret_cast<f32x4*>(yrow + 64 * j) = o0;
                        *reinterpret_cast<f32x4*>(yrow + 64 * j + 4) = o1;
                    }
                    if (srow != nullptr) {
                        *reinterpret_cast<f16x8*>(srow + 64 * j) = oh;
                        *reinterpret_cast<f16x8*>(srow + FS_HID + 64 * j) = oh;
                        *reinterpret_cast<f16x8*>(srow + 2 * FS_HID + 64 * j) = ol;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#if FS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FS_STAMP(t_end);
    if (blockIdx.x == 300 && tid == 64) {
        fs_dbg[0] = t_loop0 - t_start; fs_dbg[1] = t_loop1 - t_loop0; fs_dbg[2] = t_end - t_loop1; fs_dbg[3] = n_chunks;
    }
#endif
}

#if FS_STAMPS
}  // namespace
extern "C" int rag_enc_linear_ln_split_debug(unsigned long long* out8) {
    RAG_HIP_TRY(hipDeviceSynchronize());
    RAG_HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(fs_dbg), 8 * sizeof(unsigned long long)));
    return RAG_OK;
}
namespace {
#endif

// W [384][K] fp32 -> the two packed fp16 images the kernel streams: piece P = 1536 c + 4 f + j of an image (16
// bytes) holds columns 32 c + 8 (j ^ ((f >> 2) & 3)) + 0..7 of feature (f with bits 2 and 3 swapped).
__global__ void pack_weights_kernel(const float* __restrict__ w, int K, f16x8* __restrict__ wh, f16x8* __restrict__ wl) {
    const int n_pieces = (K / FS_C) * (FS_HID * 4);
    for (int P = blockIdx.x * blockDim.x + threadIdx.x; P < n_pieces; P += gridDim.x * blockDim.x) {
        const int c = P / (FS_HID * 4), L = P % (FS_HID * 4);
        const int f = L / 4, j = (L % 4) ^ ((f >> 2) & 3);
        const int fo = (f & ~12) | ((f & 4) << 1) | ((f & 8) >> 1);
        const float* src = w + (size_t)fo * K + FS_C * c + 8 * j;
        f16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            f16 a, b;
            split_f16(src[e], a, b);
            hi[e] = a;
            lo[e] = b;
        }
        wh[P] = hi;
        wl[P] = lo;
    }
}

bool aligned16(std::initializer_list<const void*> ptrs) {
    for (const void* q : ptrs)
        if (reinterpret_cast<uintptr_t>(q) % 16 != 0) return false;
    return true;
}

// Measurement aid: HIP event pairs around the launches (one state per process, not thread safe).
struct FsProfile {
    static constexpr int RING = 1024;
    bool on = false;
    int n = 0;
    hipEvent_t ev0[RING], ev1[RING];
    int64_t rows[RING];
    int32_t k[RING];
};
FsProfile g_fs_prof;

}  // namespace

namespace {
int linear_ln_split_launch(const void* d_in, int32_t in_is_split, const float* d_b1, const void* d_wh, const void* d_wl,
                           const float* d_b2, const float* d_residual, const void* d_residual_split, const float* d_gamma,
                           const float* d_beta, float eps, int64_t m, int32_t k, int32_t n, float* d_y, void* d_y_split,
                           void* stream_) {
    RAG_REQUIRE(n == FS_HID, RAG_ELIMIT, "rag_enc_linear_ln_split: the output width must be %d (got %d)", FS_HID, n);
    RAG_REQUIRE(k >= FS_C && k % FS_C == 0 && k <= FS_MAX_K, RAG_ELIMIT,
                "rag_enc_linear_ln_split: the input width must be a multiple of %d up to %d (got %d)", FS_C, FS_MAX_K, k);
    RAG_REQUIRE(m >= 0, RAG_EINVAL, "rag_enc_linear_ln_split: negative row count");
    RAG_REQUIRE(in_is_split >= 0 && in_is_split <= 2, RAG_EINVAL, "rag_enc_linear_ln_split: in_is_split is 0, 1 or 2 (got %d)",
                in_is_split);
    if (m == 0) return RAG_OK;
    RAG_REQUIRE(d_in && d_wh && d_wl && d_b2 && (d_residual || d_residual_split) && d_gamma && d_beta && (d_y || d_y_split) &&
                (in_is_split || d_b1), RAG_EINVAL, "rag_enc_linear_ln_split: null pointer");
    RAG_REQUIRE(d_in != (const void*)d_y && d_in != d_y_split, RAG_EINVAL, "rag_enc_linear_ln_split: outputs must not alias the input");
    RAG_REQUIRE(d_residual_split == nullptr || d_residual_split != d_y_split, RAG_EINVAL,
                "rag_enc_linear_ln_split: the split output must not alias the split residual");
    RAG_REQUIRE(aligned16({d_in, d_b1, d_wh, d_wl, d_b2, d_residual, d_residual_split, d_gamma, d_beta, d_y, d_y_split}), RAG_EINVAL,
                "rag_enc_linear_ln_split: pointers must be 16-byte aligned");
    RAG_REQUIRE(ceil_div64(m, FS_BM) < (1ll << 31), RAG_ELIMIT, "rag_enc_linear_ln_split: too many rows");
    FfnOutParams p = {};
    if (in_is_split) p.hs = (const f16*)d_in; else p.h = (const float*)d_in;
    p.hs_hl = in_is_split == 2;
    p.b1 = d_b1; p.wh = (const f16*)d_wh; p.wl = (const f16*)d_wl; p.b2 = d_b2; p.residual = d_residual;
    p.residual_s = (const f16*)d_residual_split;
    p.gamma = d_gamma; p.beta = d_beta; p.y = d_y; p.ys = (f16*)d_y_split; p.M = m; p.K = k; p.eps = eps;
    const size_t lds = (size_t)FS_OFF_B1 + 4 * ((size_t)k + 3 * FS_HID);
    hipStream_t stream = (hipStream_t)stream_;
    const dim3 grid((unsigned)ceil_div64(m, FS_BM));
    const bool timed = g_fs_prof.on && g_fs_prof.n < FsProfile::RING && !rag_stream_capturing(stream);
    if (timed) RAG_HIP_TRY(hipEventRecord(g_fs_prof.ev0[g_fs_prof.n], stream));
    if (in_is_split) {
        RAG_GRANT_LDS(ffn_out_split_kernel<1>, lds);
        hipLaunchKernelGGL(ffn_out_split_kernel<1>, grid, dim3(FS_NT), lds, stream, p);
    } else {
        RAG_GRANT_LDS(ffn_out_split_kernel<0>, lds);
        hipLaunchKernelGGL(ffn_out_split_kernel<0>, grid, dim3(FS_NT), lds, stream, p);
    }
    RAG_HIP_TRY(hipGetLastError());
    if (timed) {
        RAG_HIP_TRY(hipEventRecord(g_fs_prof.ev1[g_fs_prof.n], stream));
        g_fs_prof.rows[g_fs_prof.n] = m;
        g_fs_prof.k[g_fs_prof.n++] = k;
    }
    return RAG_OK;
}
}  // namespace

extern "C" int rag_enc_linear_ln_split_dev(const void* d_in, int32_t in_is_split, const float* d_b1, const void* d_wh,
                                           const void* d_wl, const float* d_b2, const float* d_residual,
                                           const float* d_gamma, const float* d_beta, float eps, int64_t m, int32_t k,
                                           int32_t n, float* d_y, void* d_y_split, void* stream_) {
    RAG_REQUIRE(m == 0 || d_residual, RAG_EINVAL, "rag_enc_linear_ln_split: null pointer");
    return linear_ln_split_launch(d_in, in_is_split, d_b1, d_wh, d_wl, d_b2, d_residual, nullptr, d_gamma, d_beta, eps, m, k, n,
                                  d_y, d_y_split, stream_);
}

extern "C" int rag_enc_linear_ln_split_chain_dev(const void* d_in, int32_t in_is_split, const float* d_b1, const void* d_wh,
                                                 const void* d_wl, const float* d_b2, const void* d_residual_split,
                                                 const float* d_gamma, const float* d_beta, float eps, int64_t m, int32_t k,
                                                 int32_t n, float* d_y, void* d_y_split, void* stream_) {
    RAG_REQUIRE(m == 0 || d_residual_split, RAG_EINVAL, "rag_enc_linear_ln_split_chain: null pointer");
    return linear_ln_split_launch(d_in, in_is_split, d_b1, d_wh, d_wl, d_b2, nullptr, d_residual_split, d_gamma, d_beta, eps, m,
                                  k, n, d_y, d_y_split, stream_);
}

extern "C" int rag_enc_linear_ln_split_pack_dev(const float* d_w, int32_t k, int32_t n, void* d_wh, void* d_wl, void* stream_) {
    RAG_REQUIRE(n == FS_HID, RAG_ELIMIT, "rag_enc_linear_ln_split_pack: the output width must be %d (got %d)", FS_HID, n);
    RAG_REQUIRE(k >= FS_C && k % FS_C == 0 && k <= FS_MAX_K, RAG_ELIMIT,
                "rag_enc_linear_ln_split_pack: the input width must be a multiple of %d up to %d (got %d)", FS_C, FS_MAX_K, k);
    RAG_REQUIRE(d_w && d_wh && d_wl, RAG_EINVAL, "rag_enc_linear_ln_split_pack: null pointer");
    RAG_REQUIRE(aligned16({d_w, d_wh, d_wl}), RAG_EINVAL, "rag_enc_linear_ln_split_pack: pointers must be 16-byte aligned");
    const int n_pieces = (k / FS_C) * (FS_HID * 4);
    hipLaunchKernelGGL(pack_weights_kernel, dim3((n_pieces + 255) / 256), dim3(256), 0, (hipStream_t)stream_, d_w, (int)k,
                       (f16x8*)d_wh, (f16x8*)d_wl);
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

extern "C" int rag_enc_linear_ln_split_profile(int32_t enable) {
    if (enable && !g_fs_prof.on) {
        for (int i = 0; i < FsProfile::RING; ++i) {
            RAG_HIP_TRY(hipEventCreate(&g_fs_prof.ev0[i]));
            RAG_HIP_TRY(hipEventCreate(&g_fs_prof.ev1[i]));
        }
        g_fs_prof.on = true;
    }
    if (!enable && g_fs_prof.on) {
        for (int i = 0; i < FsProfile::RING; ++i) {
            (void)hipEventDestroy(g_fs_prof.ev0[i]);
            (void)hipEventDestroy(g_fs_prof.ev1[i]);
        }
        g_fs_prof.on = false;
    }
    g_fs_prof.n = 0;
    return RAG_OK;
}

extern "C" int rag_enc_linear_ln_split_profile_read(float* out_ms, int64_t* out_rows, int32_t* out_k, int32_t max_n,
                                                    int32_t* out_n) {
    RAG_REQUIRE(out_ms && out_rows && out_k && out_n, RAG_EINVAL, "rag_enc_linear_ln_split_profile_read: null argument");
    int n = g_fs_prof.on ? g_fs_prof.n : 0;
    if (n > max_n) n = max_n;
    for (int i = 0; i < n; ++i) {
        RAG_HIP_TRY(hipEventSynchronize(g_fs_prof.ev1[i]));
        RAG_HIP_TRY(hipEventElapsedTime(&out_ms[i], g_fs_prof.ev0[i], g_fs_prof.ev1[i]));
        out_rows[i] = g_fs_prof.rows[i];
        out_k[i] = g_fs_prof.k[i];
    }
    *out_n = n;
    g_fs_prof.n = 0;
    return RAG_OK;
}
