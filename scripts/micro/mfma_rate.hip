// Issue-rate probe: v_mfma_f32_32x32x16_bf16 streams with 1 to 4 waves per SIMD and 1, 2 or 4 accumulators per
// wave (an accumulator is continued that many MFMAs after the one that wrote it).  With constant operands every
// combination reaches 2.0-2.3 PFLOP/s, one accumulator and one wave per SIMD included.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate ; run: ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void probe(float* out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    bf16x8 x, w;
    for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(seed + threadIdx.x * 0.001f + j); w[j] = (__bf16)(seed * 0.5f + j * 0.01f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int i = 0; i < 16; ++i) s += acc[a][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd, float seed) {
    const int threads = 64 * 4 * waves_per_simd, blocks = 256 * 4, iters = 4000;
    float* out;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    probe<NACC><<<blocks, threads>>>(out, 10, seed);
    hipEventRecord(a);
    probe<NACC><<<blocks, threads>>>(out, iters, seed);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double flops = 2.0 * 32 * 32 * 16 * NACC * (double)iters * (threads / 64) * blocks;
    printf("acc=%d waves/SIMD=%d seed=%g: %.3f ms, %.0f TFLOP/s\n", NACC, waves_per_simd, seed, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    for (float seed : {0.f, 1.37f}) {
        run<4>(1, seed); run<2>(1, seed); run<4>(2, seed); run<4>(4, seed);
        run<2>(2, seed); run<2>(3, seed); run<2>(4, seed); run<1>(1, seed); run<1>(3, seed);
    }
    return 0;
}
