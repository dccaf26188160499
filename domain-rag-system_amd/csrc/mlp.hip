// Feed-forward block of a 384-wide BERT layer (MiniLM-L6 cross-encoder) in ONE kernel:
//
//   Y = LayerNorm( X + GELU(X W1^T + b1) W2^T + b2 ) * gamma + beta
//
// X, Y [M][384] bf16; W1 [I][384], W2 [384][I] (nn.Linear layouts), fp32 accumulation on
// v_mfma_f32_32x32x16_bf16.  The [M][I] intermediate (I = 1536: four times the layer's input) never
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
// product if that product's k order is permuted to match, and the permutation is applied to W2
// while it is staged (no shuffles, no LDS round trip for H).  The intermediate dimension is walked
// in chunks of 32 units: per chunk 24 MFMAs for H^T, the GELU on 16 registers, 24 MFMAs into the
// twelve 32x32 tiles of Y^T (192 registers).  W1 / W2 chunks are double buffered in LDS (padded
// rows: conflict-free 16-byte reads) and fetched from L2 one chunk ahead through registers.  The
// epilogue adds bias and the residual (the block's own input), takes the LayerNorm statistics of a
// token from the lane and its partner in the other half-wave, and stores bf16.
//
// Behind CrossEncoderReranker.score_pairs (reference src/reranker/rerank.py:79-84): the
// intermediate / output sub-layers of the BertForSequenceClassification forward that
// CrossEncoder.predict runs.
#include "common.h"

#include <math.h>

namespace {

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int ML_HID = 384;                  // layer width
constexpr int ML_BM = 128;                   // tokens per workgroup
constexpr int ML_NT = 256;                   // four waves
constexpr int ML_C = 32;                     // intermediate units per chunk
constexpr int ML_KS = ML_HID / 16;           // k-steps of the first product: 24
constexpr int ML_YT = ML_HID / 32;           // accumulator tiles of Y^T: 12
constexpr int ML_W_BYTES = ML_C * ML_HID * 2;             // one W1 chunk = one W2 chunk = 24576 B
constexpr int ML_W_DMAS = ML_W_BYTES / 1024 / (ML_NT / 64);   // 1-KiB direct-to-LDS loads per wave and chunk: 6
constexpr int ML_PF = 4;                    // fragment reads in flight ahead of the MFMAs, in k-steps
constexpr int ML_MAX_I = 8192;               // b1 sits in LDS
// LDS image: W1 buffers [2][32 rows][48 x 16 B], W2 buffers [2][384 rows][4 x 16 B], b1 [I]
constexpr int ML_OFF_W2 = 2 * ML_W_BYTES;
constexpr int ML_OFF_B1 = 4 * ML_W_BYTES;

struct MlpParams {
    const bf16* x;        // [M][384]
    const bf16* w1;       // [I][384]
    const bf16* b1;       // [I]
    const bf16* w2;       // [384][I]
    const bf16* b2;       // [384]
    const bf16* gamma;    // [384]
    const bf16* beta;     // [384]
    bf16* y;              // [M][384]
    int64_t M;
    int I;
    float eps;
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

// GELU(v) = v/2 (1 + erf(v / sqrt 2)) for two values at a time (packed fp32 FMAs).  erf(z) is an odd
// degree-17 polynomial fitted on |z| <= 3.2 and held constant beyond (erf(3.2) = 1 - 6e-6): absolute
// error of the GELU below 4e-5, a fiftieth of the bf16 rounding of values near 1 — and no
// exponential or reciprocal, which run at a quarter of the FMA rate.
__device__ __forceinline__ f32x2 gelu2(f32x2 v) {
    f32x2 z = v * 0.70710678118654752f;
    z.x = fminf(fmaxf(z.x, -3.2f), 3.2f);
    z.y = fminf(fmaxf(z.y, -3.2f), 3.2f);
    const f32x2 s = z * z;
    f32x2 p = {2.086896185e-08f, 2.086896185e-08f};
    p = __builtin_elementwise_fma(p, s, (f32x2){-1.152973149e-06f, -1.152973149e-06f});
    p = __builtin_elementwise_fma(p, s, (f32x2){2.798506678e-05f, 2.798506678e-05f});
    p = __builtin_elementwise_fma(p, s, (f32x2){-3.954573252e-04f, -3.954573252e-04f});
    p = __builtin_elementwise_fma(p, s, (f32x2){3.646156001e-03f, 3.646156001e-03f});
    p = __builtin_elementwise_fma(p, s, (f32x2){-2.336560723e-02f, -2.336560723e-02f});
    p = __builtin_elementwise_fma(p, s, (f32x2){1.082771112e-01f, 1.082771112e-01f});
    p = __builtin_elementwise_fma(p, s, (f32x2){-3.731665065e-01f, -3.731665065e-01f});
    p = __builtin_elementwise_fma(p, s, (f32x2){1.127723030e+00f, 1.127723030e+00f});
    const f32x2 e = p * z;                                    // erf(v / sqrt 2)
    const f32x2 hv = v * 0.5f;
    return __builtin_elementwise_fma(hv, e, hv);
}

__global__ __launch_bounds__(ML_NT, 1) void mlp_fused_kernel(MlpParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
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
    //   W2 image row f (feature) holds the chunk's 32 units, piece j at j ^ ((f >> 2) & 3).
    uint32_t off1[ML_W_DMAS], off2[ML_W_DMAS];                // per-lane source byte offsets inside a chunk
#pragma unroll
    for (int i = 0; i < ML_W_DMAS; ++i) {
        const int L = 64 * (wave + 4 * i) + lane;             // piece index inside the image
        const int q = L / 48, j1 = (L % 48) ^ (q & 15);
        const int unit = (q & ~12) | ((q & 4) << 1) | ((q & 8) >> 1);
        off1[i] = (uint32_t)((unit * ML_HID + j1 * 8) * 2);
        const int f = L / 4, j2 = (L % 4) ^ ((f >> 2) & 3);
        off2[i] = (uint32_t)((f * p.I + j2 * 8) * 2);
    }
    auto dma_w1 = [&](int c, int buf) {                      // chunk c of W1 -> W1 buffer buf
        const bf16* g = p.w1 + (size_t)c * ML_C * ML_HID;
#pragma unroll
        for (int i = 0; i < ML_W_DMAS; ++i) lds_dma16(lds0 + buf * ML_W_BYTES + 1024 * (wave + 4 * i), g, off1[i]);
    };
    auto dma_w2 = [&](int c, int buf) {
        const bf16* g = p.w2 + c * ML_C;
#pragma unroll
        for (int i = 0; i < ML_W_DMAS; ++i)
            lds_dma16(lds0 + ML_OFF_W2 + buf * ML_W_BYTES + 1024 * (wave + 4 * i), g, off2[i]);
    };
    const int last = n_chunks - 1;
    dma_w1(0, 0);
    dma_w1(min(1, last), 1);
    dma_w2(0, 0);
    for (int i = tid; i < p.I / 8; i += ML_NT)
        *reinterpret_cast<bf16x8*>(smem + ML_OFF_B1 + 16 * i) = *reinterpret_cast<const bf16x8*>(p.b1 + 8 * i);

    // ---- the wave's 32 rows of X as B fragments: lane (r, h) holds X[tok][16t + 8h .. + 7]
    bf16x8 xf[ML_KS];
    {
        const bf16* xrow = p.x + tok_ld * ML_HID + 8 * h;
#pragma unroll
        for (int t = 0; t < ML_KS; ++t) xf[t] = *reinterpret_cast<const bf16x8*>(xrow + 16 * t);
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
    const unsigned char* b1s = smem + ML_OFF_B1 + 16 * h;

    auto bias_init = [&](int c, f32x16& acc) {               // registers 8t..8t+7 = units 16t + 8h + 0..7
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(b1s + c * (ML_C * 2) + 32 * t);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[8 * t + j] = (float)b[j];
        }
    };
    auto gemm1 = [&](int buf, f32x16& acc) {
#pragma unroll
        for (int t = 0; t < ML_KS; ++t) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(a1[t & 7] + buf * ML_W_BYTES + 256 * (t >> 3));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xf[t], acc, 0, 0, 0);
        }
    };
    auto gelu_pack = [&](const f32x16& acc, bf16x8* pf) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const f32x2 g = gelu2((f32x2){acc[i], acc[i + 1]});
            pf[i >> 3][i & 7] = (bf16)g.x;
            pf[i >> 3][(i & 7) + 1] = (bf16)g.y;
        }
    };

    f32x16 yacc[ML_YT];
#pragma unroll
    for (int m = 0; m < ML_YT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) yacc[m][i] = 0.f;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- prologue: H^T of chunks 0 and 1; then W1 buffer 0 is free for chunk 2
    f32x16 hraw, hacc;
    bf16x8 pf[2];
    bias_init(0, hacc);
    gemm1(0, hacc);
    bias_init(min(1, last), hraw);
    gemm1(1, hraw);
    __syncthreads();
    dma_w1(min(2, last), 0);
    gelu_pack(hacc, pf);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- main loop, three chunks in flight: iteration c runs H^T of chunk c+2 (into hacc), the GELU
    // of chunk c+1 (hraw -> pfn) and Y^T += W2 . H^T of chunk c (pf) — independent of one another, so
    // the matrix unit, the vector ALU and the LDS reads overlap — while chunks c+3 of W1 and c+1 of W2
    // stream into the buffers the previous iteration finished with.  Past the last chunk the loads
    // and the first product repeat the last chunk (results unused): no branches in the loop body.
    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        dma_w1(min(c + 3, last), buf ^ 1);
        dma_w2(min(c + 1, last), buf ^ 1);
        bias_init(min(c + 2, last), hacc);
        bf16x8 pfn[2];
        const unsigned char* w2b0 = a2[0] + buf * ML_W_BYTES;
        const unsigned char* w2b1 = a2[1] + buf * ML_W_BYTES;
        // fragment reads run ML_PF steps ahead of the MFMAs that consume them
        auto frag1 = [&](int s) {
            return *reinterpret_cast<const bf16x8*>(a1[s & 7] + buf * ML_W_BYTES + 256 * (s >> 3));
        };
        auto frag2 = [&](int s) {
            return *reinterpret_cast<const bf16x8*>((s / ML_YT ? w2b1 : w2b0) + (s % ML_YT) * (32 * ML_C * 2));
        };
        bf16x8 q1[ML_PF], q2[ML_PF];
#pragma unroll
        for (int s = 0; s < ML_PF - 1; ++s) { q1[s] = frag1(s); q2[s] = frag2(s); }
#pragma unroll
        for (int s = 0; s < ML_KS; ++s) {
            if (s + ML_PF - 1 < ML_KS) {
                q1[(s + ML_PF - 1) % ML_PF] = frag1(s + ML_PF - 1);
                q2[(s + ML_PF - 1) % ML_PF] = frag2(s + ML_PF - 1);
            }
            hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q1[s % ML_PF], xf[s], hacc, 0, 0, 0);
            const int m = s % ML_YT, t = s / ML_YT;
            yacc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q2[s % ML_PF], pf[t], yacc[m], 0, 0, 0);
            if (s % 3 == 0) {
                const int i = 2 * (s / 3);
                const f32x2 g = gelu2((f32x2){hraw[i], hraw[i + 1]});
                pfn[i >> 3][i & 7] = (bf16)g.x;
                pfn[i >> 3][(i & 7) + 1] = (bf16)g.y;
            }
        }
        pf[0] = pfn[0];
        pf[1] = pfn[1];
        hraw = hacc;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: yacc[m][i] is feature 32m + (i&3) + 8(i>>2) + 4h of token `tok`
    const bf16* xres = p.x + tok_ld * ML_HID;
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < ML_YT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = 32 * m + 8 * q + 4 * h;
            const bf16x4 bb = *reinterpret_cast<const bf16x4*>(p.b2 + n);
            const bf16x4 rr = *reinterpret_cast<const bf16x4*>(xres + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = yacc[m][4 * q + j] + (float)bb[j] + (float)rr[j];
                yacc[m][4 * q + j] = v;
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
    for (int m = 0; m < ML_YT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = 32 * m + 8 * q + 4 * h;
            const bf16x4 g4 = *reinterpret_cast<const bf16x4*>(p.gamma + n);
            const bf16x4 be4 = *reinterpret_cast<const bf16x4*>(p.beta + n);
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (bf16)((yacc[m][4 * q + j] - mean) * rstd * (float)g4[j] + (float)be4[j]);
            if (live) *reinterpret_cast<bf16x4*>(p.y + tok * ML_HID + n) = o;
        }
}

}  // namespace

extern "C" int rag_enc_mlp_dev(const void* d_x, const void* d_w1, const void* d_b1, const void* d_w2, const void* d_b2,
                               const void* d_gamma, const void* d_beta, float eps, int64_t m, int32_t hidden,
                               int32_t intermediate, int32_t dtype, void* d_y, void* stream_) {
    RAG_REQUIRE(dtype == RAG_DTYPE_BF16, RAG_ELIMIT, "rag_enc_mlp: bf16 only");
    RAG_REQUIRE(hidden == ML_HID, RAG_ELIMIT, "rag_enc_mlp: hidden size must be %d (got %d)", ML_HID, hidden);
    RAG_REQUIRE(intermediate >= ML_C && intermediate % ML_C == 0 && intermediate <= ML_MAX_I, RAG_ELIMIT,
                "rag_enc_mlp: intermediate size must be a multiple of %d up to %d (got %d)", ML_C, ML_MAX_I, intermediate);
    RAG_REQUIRE(m >= 0, RAG_EINVAL, "rag_enc_mlp: negative row count");
    if (m == 0) return RAG_OK;
    RAG_REQUIRE(d_x && d_w1 && d_b1 && d_w2 && d_b2 && d_gamma && d_beta && d_y, RAG_EINVAL, "rag_enc_mlp: null pointer");
    RAG_REQUIRE(d_x != d_y, RAG_EINVAL, "rag_enc_mlp: y must not alias x (x is read again as the residual)");
    RAG_REQUIRE(ceil_div64(m, ML_BM) < (1ll << 31), RAG_ELIMIT, "rag_enc_mlp: too many rows");
    const size_t lds_bytes = (size_t)ML_OFF_B1 + 2 * (size_t)intermediate;
    static bool attr_set = false;
    if (!attr_set) {
        RAG_HIP_TRY(hipFuncSetAttribute((const void*)mlp_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        ML_OFF_B1 + 2 * ML_MAX_I));
        attr_set = true;
    }
    MlpParams p;
    p.x = (const bf16*)d_x; p.w1 = (const bf16*)d_w1; p.b1 = (const bf16*)d_b1; p.w2 = (const bf16*)d_w2;
    p.b2 = (const bf16*)d_b2; p.gamma = (const bf16*)d_gamma; p.beta = (const bf16*)d_beta; p.y = (bf16*)d_y;
    p.M = m; p.I = intermediate; p.eps = eps;
    hipLaunchKernelGGL(mlp_fused_kernel, dim3((unsigned)ceil_div64(m, ML_BM)), dim3(ML_NT), lds_bytes,
                       (hipStream_t)stream_, p);
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}
