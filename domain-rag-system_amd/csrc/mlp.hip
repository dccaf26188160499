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
constexpr int ML_LD1 = ML_HID + 8;           // W1 chunk row in LDS, elements (784 B)
constexpr int ML_LD2 = ML_C + 8;             // W2 chunk row in LDS, elements (80 B)
constexpr int ML_KS = ML_HID / 16;           // k-steps of the first product: 24
constexpr int ML_YT = ML_HID / 32;           // accumulator tiles of Y^T: 12
constexpr int ML_W1_PIECES = ML_C * (ML_HID / 8) / ML_NT;   // 16-byte pieces per thread and chunk: 6
constexpr int ML_W2_PIECES = ML_HID * (ML_C / 8) / ML_NT;   // 6
constexpr int ML_STAGE_ELEMS = ML_C * ML_LD1 + ML_HID * ML_LD2;
constexpr size_t ML_LDS_BYTES = (size_t)2 * ML_STAGE_ELEMS * sizeof(bf16);

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
    bf16* const lds = reinterpret_cast<bf16*>(smem);          // [2] x { W1 chunk [32][392], W2 chunk [384][40] }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * ML_BM;
    const int64_t tok = m0 + wave * 32 + r;                   // this lane's token
    const int64_t tok_ld = min(tok, p.M - 1);                 // rows past M read the last row; never stored
    const int n_chunks = p.I / ML_C;

    // ---- the wave's 32 rows of X as B fragments: lane (r, h) holds X[tok][16t + 8h .. + 7]
    bf16x8 xf[ML_KS];
    {
        const bf16* xrow = p.x + tok_ld * ML_HID + 8 * h;
#pragma unroll
        for (int t = 0; t < ML_KS; ++t) xf[t] = *reinterpret_cast<const bf16x8*>(xrow + 16 * t);
    }

    // ---- weight staging: global -> registers one chunk ahead -> LDS
    bf16x8 s1[ML_W1_PIECES], s2[ML_W2_PIECES];
    bf16x4 b1n[4], b1c[4];
    auto chunk_load = [&](int c) {
#pragma unroll
        for (int i = 0; i < ML_W1_PIECES; ++i) {
            const int idx = tid + i * ML_NT, row = idx / (ML_HID / 8), pc = idx % (ML_HID / 8);
            s1[i] = *reinterpret_cast<const bf16x8*>(p.w1 + (size_t)(c * ML_C + row) * ML_HID + pc * 8);
        }
#pragma unroll
        for (int i = 0; i < ML_W2_PIECES; ++i) {
            const int idx = tid + i * ML_NT, feat = idx / (ML_C / 8), pc = idx % (ML_C / 8);
            s2[i] = *reinterpret_cast<const bf16x8*>(p.w2 + (size_t)feat * p.I + c * ML_C + pc * 8);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) b1n[q] = *reinterpret_cast<const bf16x4*>(p.b1 + c * ML_C + 8 * q + 4 * h);
    };
    auto chunk_store = [&](int buf) {
        bf16* w1s = lds + buf * ML_STAGE_ELEMS;
        bf16* w2s = w1s + ML_C * ML_LD1;
#pragma unroll
        for (int i = 0; i < ML_W1_PIECES; ++i) {
            const int idx = tid + i * ML_NT, row = idx / (ML_HID / 8), pc = idx % (ML_HID / 8);
            *reinterpret_cast<bf16x8*>(w1s + row * ML_LD1 + pc * 8) = s1[i];
        }
#pragma unroll
        for (int i = 0; i < ML_W2_PIECES; ++i) {
            const int idx = tid + i * ML_NT, feat = idx / (ML_C / 8), pc = idx % (ML_C / 8);
            // units 8pc .. 8pc+7 of the chunk; inside a group of 16 units, unit 4a + b (a < 4, b < 4)
            // goes to k position 8(a&1) + 4(a>>1) + b: the order in which the H^T accumulators hold them
            const int g = pc >> 1, half = pc & 1;
            bf16x4 lo, hi;
#pragma unroll
            for (int j = 0; j < 4; ++j) { lo[j] = s2[i][j]; hi[j] = s2[i][4 + j]; }
            *reinterpret_cast<bf16x4*>(w2s + feat * ML_LD2 + 16 * g + 4 * half) = lo;
            *reinterpret_cast<bf16x4*>(w2s + feat * ML_LD2 + 16 * g + 8 + 4 * half) = hi;
        }
    };

    f32x16 yacc[ML_YT];
#pragma unroll
    for (int m = 0; m < ML_YT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) yacc[m][i] = 0.f;

    chunk_load(0);
    chunk_store(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) b1c[q] = b1n[q];
    chunk_load(min(1, n_chunks - 1));
    __syncthreads();

    for (int c = 0; c < n_chunks; ++c) {
        const int buf = c & 1;
        const bf16* w1s = lds + buf * ML_STAGE_ELEMS;
        const bf16* w2s = w1s + ML_C * ML_LD1;

        // chunk c+1 (in registers since the last iteration) goes to the other buffer, whose readers
        // all passed the barrier that ended the last iteration; chunk c+2 takes the registers.  Past the
        // last chunk the loads repeat it: no branch in the load stream.
        chunk_store(buf ^ 1);
        bf16x4 b1k[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b1k[q] = b1c[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) b1c[q] = b1n[q];
        chunk_load(min(c + 2, n_chunks - 1));

        // ---- H^T = W1c . X^T: two accumulators (even / odd k-steps) so consecutive MFMAs are independent
        f32x16 h0, h1;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            h0[i] = (float)b1k[i >> 2][i & 3];                 // row (i&3) + 8(i>>2) + 4h of the chunk
            h1[i] = 0.f;
        }
        {
            const bf16* a = w1s + r * ML_LD1 + 8 * h;
#pragma unroll
            for (int t = 0; t < ML_KS; t += 2) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(a + 16 * t);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(a + 16 * t + 16);
                h0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, xf[t], h0, 0, 0, 0);
                h1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xf[t + 1], h1, 0, 0, 0);
            }
        }
        // ---- GELU, then the accumulators become the B fragments of the second product
        bf16x8 pf[2];
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const f32x2 g = gelu2((f32x2){h0[i] + h1[i], h0[i + 1] + h1[i + 1]});
            pf[i >> 3][i & 7] = (bf16)g.x;
            pf[i >> 3][(i & 7) + 1] = (bf16)g.y;
        }
        // ---- Y^T += W2c . H^T
        {
            const bf16* a = w2s + r * ML_LD2 + 8 * h;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int m = 0; m < ML_YT; ++m) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(a + m * 32 * ML_LD2 + 16 * t);
                    yacc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, pf[t], yacc[m], 0, 0, 0);
                }
        }
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
    RAG_REQUIRE(intermediate >= ML_C && intermediate % ML_C == 0, RAG_ELIMIT,
                "rag_enc_mlp: intermediate size must be a multiple of %d (got %d)", ML_C, intermediate);
    RAG_REQUIRE(m >= 0, RAG_EINVAL, "rag_enc_mlp: negative row count");
    if (m == 0) return RAG_OK;
    RAG_REQUIRE(d_x && d_w1 && d_b1 && d_w2 && d_b2 && d_gamma && d_beta && d_y, RAG_EINVAL, "rag_enc_mlp: null pointer");
    RAG_REQUIRE(d_x != d_y, RAG_EINVAL, "rag_enc_mlp: y must not alias x (x is read again as the residual)");
    RAG_REQUIRE(ceil_div64(m, ML_BM) < (1ll << 31), RAG_ELIMIT, "rag_enc_mlp: too many rows");
    static bool attr_set = false;
    if (!attr_set) {
        RAG_HIP_TRY(hipFuncSetAttribute((const void*)mlp_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)ML_LDS_BYTES));
        attr_set = true;
    }
    MlpParams p;
    p.x = (const bf16*)d_x; p.w1 = (const bf16*)d_w1; p.b1 = (const bf16*)d_b1; p.w2 = (const bf16*)d_w2;
    p.b2 = (const bf16*)d_b2; p.gamma = (const bf16*)d_gamma; p.beta = (const bf16*)d_beta; p.y = (bf16*)d_y;
    p.M = m; p.I = intermediate; p.eps = eps;
    hipLaunchKernelGGL(mlp_fused_kernel, dim3((unsigned)ceil_div64(m, ML_BM)), dim3(ML_NT), ML_LDS_BYTES,
                       (hipStream_t)stream_, p);
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}
