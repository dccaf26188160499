// Fused encoder GEMMs for 384-wide BERT layers (MiniLM-L6 cross-encoder): bf16 inputs, fp32
// accumulation on v_mfma_f32_32x32x16_bf16, and the layer's elementwise tail in the epilogue,
// so that the activation never makes an extra trip through HBM between the GEMM and its
// bias / GELU / residual / LayerNorm.  gfx950 only.
//
//   Y[m][n] = epilogue( sum_k X[m][k] * W[n][k] + bias[n] ),   X [M][K], W [N][K] (nn.Linear layout)
//
// One workgroup (8 waves) owns 128 rows of X and a block of 384 output columns; wave (g, c) has
// rows 32g..32g+31 and columns 192c..192c+191.  With the whole 384-wide output row inside one
// workgroup the LayerNorm statistics need one small exchange through LDS between the two waves
// of a row group.  The product is computed transposed (W is the MFMA's A operand, X^T its B
// operand): a lane then holds ONE output row m (its column index) and 6 x 16 values of that row
// along n in its accumulators, so bias, residual, mean and variance are register arithmetic
// plus one exchange between the two lane halves.  k-slices of 64 are staged through double-buffered LDS
// (144-byte padded rows: conflict-free 16-byte operand reads), global loads run one stage ahead
// in registers.
//
// Behind CrossEncoderReranker.score_pairs (reference src/reranker/rerank.py:79-84): these are
// the dense layers of the BertForSequenceClassification forward that CrossEncoder.predict runs.
#include "common.h"

#include <math.h>

namespace {

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GM_BM = 128;            // rows of X per workgroup
constexpr int GM_BN = 384;            // output columns per workgroup (= hidden size: full rows for LayerNorm)
constexpr int GM_BK = 64;             // k-slice per stage
constexpr int GM_LDK = GM_BK + 8;     // padded LDS row, elements (144 bytes)
constexpr int GM_NT = 512;            // 8 waves: 4 row groups x 2 column halves
constexpr int GM_TILES = GM_BN / 64;  // 6 accumulator tiles (32 x 32) per wave
constexpr int GM_X_LOADS = GM_BM * (GM_BK / 8) / GM_NT;   // 16-byte chunks per thread per stage: 1
constexpr int GM_W_LOADS = GM_BN * (GM_BK / 8) / GM_NT;   // 3
constexpr size_t GM_LDS_BYTES = (size_t)2 * (GM_BM + GM_BN) * GM_LDK * sizeof(bf16) + GM_BM * 2 * sizeof(float);

enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RESIDUAL_LN = 2 };

// GELU(v) = v/2 (1 + erf(v / sqrt 2)) with erf from Abramowitz & Stegun 7.1.26 (absolute error
// below 1.5e-7, two orders of magnitude under the bf16 rounding of the result); branch free and
// short, so 96 of them per lane fit the register budget of the epilogue.
__device__ __forceinline__ float gelu_erf(float v) {
    const float x = fabsf(v) * 0.70710678118654752f;
    const float t = __frcp_rn(1.0f + 0.3275911f * x);
    float poly = 1.061405429f;
    poly = poly * t - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    const float e = 1.0f - poly * t * __expf(-x * x);          // erf(|v| / sqrt 2)
    return 0.5f * v + 0.5f * fabsf(v) * e;                     // v/2 (1 + sign(v) e)
}

struct GemmParams {
    const bf16* x;        // [M][K]
    const bf16* w;        // [N][K]
    const bf16* bias;     // [N]
    const bf16* residual; // [M][N]   (EPI_RESIDUAL_LN)
    const bf16* gamma;    // [N]
    const bf16* beta;     // [N]
    bf16* y;              // [M][N]
    int64_t M;
    int K, N;
    float eps;
};

template <int EPI>
__global__ __launch_bounds__(GM_NT, 1) void encoder_gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16* Xs = reinterpret_cast<bf16*>(smem);                 // [2][GM_BM][GM_LDK]
    bf16* Ws = Xs + 2 * GM_BM * GM_LDK;                       // [2][GM_BN][GM_LDK]
    float* stat = reinterpret_cast<float*>(Ws + 2 * GM_BN * GM_LDK);   // [GM_BM][2] row sums of the two column halves
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int rg = wave >> 1, ch = wave & 1;                  // row group, column half
    const int64_t m0 = (int64_t)blockIdx.x * GM_BM;
    const int n0 = blockIdx.y * GM_BN;
    const int n_stages = p.K / GM_BK;

    // Global loads run TWO stages ahead of the MFMAs in two register sets (a stage of bf16 MFMAs
    // is shorter than a round trip to memory), the set of stage s+1 is parked in LDS while
    // stage s is consumed.
    bf16x8 xr[2][GM_X_LOADS], wr[2][GM_W_LOADS];
    auto stage_load = [&](int st, bf16x8* xs, bf16x8* ws) {
        const int k0 = st * GM_BK;
#pragma unroll
        for (int i = 0; i < GM_X_LOADS; ++i) {
            const int idx = tid + i * GM_NT, row = idx / (GM_BK / 8), c = idx % (GM_BK / 8);
            // rows past M read row M-1 instead (their results are never stored): no branch in the
            // load stream, so the compiler can count outstanding loads exactly
            const int64_t mr = min(m0 + row, p.M - 1);
            xs[i] = *reinterpret_cast<const bf16x8*>(p.x + mr * p.K + k0 + c * 8);
        }
#pragma unroll
        for (int i = 0; i < GM_W_LOADS; ++i) {
            const int idx = tid + i * GM_NT, row = idx / (GM_BK / 8), c = idx % (GM_BK / 8);
            ws[i] = *reinterpret_cast<const bf16x8*>(p.w + (size_t)(n0 + row) * p.K + k0 + c * 8);
        }
    };
    auto stage_store = [&](int buf, const bf16x8* xs, const bf16x8* ws) {
        bf16* xb = Xs + buf * GM_BM * GM_LDK;
        bf16* wb = Ws + buf * GM_BN * GM_LDK;
#pragma unroll
        for (int i = 0; i < GM_X_LOADS; ++i) {
            const int idx = tid + i * GM_NT, row = idx / (GM_BK / 8), c = idx % (GM_BK / 8);
            *reinterpret_cast<bf16x8*>(xb + row * GM_LDK + c * 8) = xs[i];
        }
#pragma unroll
        for (int i = 0; i < GM_W_LOADS; ++i) {
            const int idx = tid + i * GM_NT, row = idx / (GM_BK / 8), c = idx % (GM_BK / 8);
            *reinterpret_cast<bf16x8*>(wb + row * GM_LDK + c * 8) = ws[i];
        }
    };

    f32x16 acc[GM_TILES];
#pragma unroll
    for (int t = 0; t < GM_TILES; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    auto compute = [&](int buf) {
        const bf16* xb = Xs + buf * GM_BM * GM_LDK + (rg * 32 + r) * GM_LDK + 8 * h;
        const bf16* wb = Ws + buf * GM_BN * GM_LDK + (ch * 192 + r) * GM_LDK + 8 * h;
#pragma unroll
        for (int kk = 0; kk < GM_BK / 16; ++kk) {
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xb + 16 * kk);           // X^T fragment (B operand)
#pragma unroll
            for (int t = 0; t < GM_TILES; ++t) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wb + t * 32 * GM_LDK + 16 * kk);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[t], 0, 0, 0);
            }
        }
    };

    stage_load(0, xr[0], wr[0]);
    if (n_stages > 1) stage_load(1, xr[1], wr[1]);
    stage_store(0, xr[0], wr[0]);
    if (n_stages > 2) stage_load(2, xr[0], wr[0]);
    __syncthreads();
    // stage st sits in LDS buffer st & 1; stage st+1 in register set (st+1) & 1; stage st+2 in the other set
    for (int st = 0; st < n_stages; st += 2) {
        compute(0);
        if (st + 1 < n_stages) stage_store(1, xr[1], wr[1]);
        if (st + 3 < n_stages) stage_load(st + 3, xr[1], wr[1]);
        __syncthreads();
        if (st + 1 < n_stages) {
            compute(1);
            if (st + 2 < n_stages) stage_store(0, xr[0], wr[0]);
            if (st + 4 < n_stages) stage_load(st + 4, xr[0], wr[0]);
            __syncthreads();
        }
    }

    // ---- epilogue: this lane's output row is m; acc[t][i] is column nb + 32t + (i&3) + 8(i>>2) + 4h
    const int64_t m = m0 + rg * 32 + r;
    const int nb = n0 + ch * 192;
    const bool live = m < p.M;
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < GM_TILES; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = nb + 32 * t + 8 * g + 4 * h;
            const bf16x4 b4 = *reinterpret_cast<const bf16x4*>(p.bias + n);
            bf16x4 r4;
#pragma unroll
            for (int j = 0; j < 4; ++j) r4[j] = (bf16)0.f;
            if (EPI == EPI_RESIDUAL_LN && live) r4 = *reinterpret_cast<const bf16x4*>(p.residual + m * p.N + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[t][4 * g + j] + (float)b4[j];
                if (EPI == EPI_GELU) v = gelu_erf(v);
                if (EPI == EPI_RESIDUAL_LN) {
                    v += (float)r4[j];
                    sum += v;
                }
                acc[t][4 * g + j] = v;
            }
            if (EPI == EPI_GELU) __builtin_amdgcn_sched_barrier(0);   // erf is long: four values at a time
        }
        __builtin_amdgcn_sched_barrier(0);          // one tile at a time: keeps the live temporaries few
    }
    if (EPI == EPI_RESIDUAL_LN) {
        // row statistics: lane halves first, then the partner wave's column half through LDS
        sum += __shfl_xor(sum, 32, 64);
        if (h == 0) stat[(rg * 32 + r) * 2 + ch] = sum;
        __syncthreads();
        const float mean = (stat[(rg * 32 + r) * 2] + stat[(rg * 32 + r) * 2 + 1]) / (float)GM_BN;
        __syncthreads();
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < GM_TILES; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) { const float d = acc[t][i] - mean; sq += d * d; }
        sq += __shfl_xor(sq, 32, 64);
        if (h == 0) stat[(rg * 32 + r) * 2 + ch] = sq;
        __syncthreads();
        const float rstd = rsqrtf((stat[(rg * 32 + r) * 2] + stat[(rg * 32 + r) * 2 + 1]) / (float)GM_BN + p.eps);
#pragma unroll
        for (int t = 0; t < GM_TILES; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nb + 32 * t + 8 * g + 4 * h;
                const bf16x4 g4 = *reinterpret_cast<const bf16x4*>(p.gamma + n);
                const bf16x4 be4 = *reinterpret_cast<const bf16x4*>(p.beta + n);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[t][4 * g + j] = (acc[t][4 * g + j] - mean) * rstd * (float)g4[j] + (float)be4[j];
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    if (live) {
#pragma unroll
        for (int t = 0; t < GM_TILES; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (bf16)acc[t][4 * g + j];
                *reinterpret_cast<bf16x4*>(p.y + m * p.N + nb + 32 * t + 8 * g + 4 * h) = v;
                __builtin_amdgcn_sched_barrier(0);
            }
    }
}

template <int EPI>
int launch_gemm(const GemmParams& p, hipStream_t stream) {
    RAG_GRANT_LDS(encoder_gemm_kernel<EPI>, GM_LDS_BYTES);
    const dim3 grid((unsigned)ceil_div64(p.M, GM_BM), (unsigned)(p.N / GM_BN));
    hipLaunchKernelGGL(encoder_gemm_kernel<EPI>, grid, dim3(GM_NT), GM_LDS_BYTES, stream, p);
    RAG_HIP_TRY(hipGetLastError());
    return RAG_OK;
}

}  // namespace

extern "C" int rag_enc_linear_dev(const void* d_x, const void* d_w, const void* d_bias, int64_t m, int32_t k, int32_t n,
                                  int32_t epilogue, const void* d_residual, const void* d_gamma, const void* d_beta,
                                  float eps, int32_t dtype, void* d_y, void* stream_) {
    RAG_REQUIRE(dtype == RAG_DTYPE_BF16, RAG_ELIMIT, "rag_enc_linear: bf16 only");
    RAG_REQUIRE(m >= 0 && k >= GM_BK && k % GM_BK == 0 && n >= GM_BN && n % GM_BN == 0, RAG_ELIMIT,
                "rag_enc_linear: need k %% %d == 0 and n %% %d == 0 (got k=%d, n=%d)", GM_BK, GM_BN, k, n);
    RAG_REQUIRE(epilogue == RAG_EPILOGUE_BIAS || epilogue == RAG_EPILOGUE_BIAS_GELU ||
                    epilogue == RAG_EPILOGUE_BIAS_RESIDUAL_LAYERNORM,
                RAG_EINVAL, "rag_enc_linear: unknown epilogue %d", epilogue);
    RAG_REQUIRE(epilogue != RAG_EPILOGUE_BIAS_RESIDUAL_LAYERNORM || n == GM_BN, RAG_ELIMIT,
                "rag_enc_linear: the LayerNorm epilogue needs n == %d", GM_BN);
    if (m == 0) return RAG_OK;
    RAG_REQUIRE(d_x && d_w && d_bias && d_y, RAG_EINVAL, "rag_enc_linear: null pointer");
    RAG_REQUIRE(epilogue != RAG_EPILOGUE_BIAS_RESIDUAL_LAYERNORM || (d_residual && d_gamma && d_beta), RAG_EINVAL,
                "rag_enc_linear: the LayerNorm epilogue needs residual, gamma and beta");
    RAG_REQUIRE(ceil_div64(m, GM_BM) < (1ll << 31), RAG_ELIMIT, "rag_enc_linear: too many rows");
    GemmParams p;
    p.x = (const bf16*)d_x; p.w = (const bf16*)d_w; p.bias = (const bf16*)d_bias;
    p.residual = (const bf16*)d_residual; p.gamma = (const bf16*)d_gamma; p.beta = (const bf16*)d_beta;
    p.y = (bf16*)d_y; p.M = m; p.K = k; p.N = n; p.eps = eps;
    hipStream_t stream = (hipStream_t)stream_;
    if (epilogue == RAG_EPILOGUE_BIAS) return launch_gemm<EPI_BIAS>(p, stream);
    if (epilogue == RAG_EPILOGUE_BIAS_GELU) return launch_gemm<EPI_GELU>(p, stream);
    return launch_gemm<EPI_RESIDUAL_LN>(p, stream);
}
