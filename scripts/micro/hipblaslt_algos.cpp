// Every hipBLASLt solution for the cross-encoder's fp16-in / fp32-out products, timed on the device.
//
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/hipblaslt_algos.cpp -o gpurun_out/hipblaslt_algos -lhipblaslt
//   ./hipblaslt_algos [tokens] [top]
//
// Row-major view: out[T, n] (row stride ldo) = x[T, k] . w[n, k]^T, x and w fp16, out fp32.  hipBLASLt is column-major:
// D'[n, T] = op(A') op(B') with A' = w read as [k, n] (ld k, transposed) and B' = x read as [k, T] (ld k).
// Prints, per shape, the library's own first choice and the fastest `top` solutions with their indices.
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-ext.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define CB(x) do { hipblasStatus_t s_ = (x); if (s_ != HIPBLAS_STATUS_SUCCESS) { fprintf(stderr, "%s: status %d\n", #x, (int)s_); exit(2); } } while (0)

__global__ void fill_half(_Float16* p, size_t n, unsigned seed)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f));
    }
}

struct Shape { int n, k, ldo; const char* what; };

int main(int argc, char** argv)
{
    long tokens = argc > 1 ? atol(argv[1]) : 688128;
    int top = argc > 2 ? atoi(argv[2]) : 8;
    const int kmax = 1152, nmax = 1536;
    _Float16 *x[2], *w; float* out[2]; void* ws; size_t ws_bytes = 128u << 20;
    for (int i = 0; i < 2; ++i) {
        CK(hipMalloc(&x[i], (size_t)tokens * kmax * 2));
        CK(hipMalloc(&out[i], (size_t)tokens * nmax * 4));
        fill_half<<<4096, 256>>>(x[i], (size_t)tokens * kmax, 17 + i);
    }
    CK(hipMalloc(&w, (size_t)nmax * kmax * 2));
    fill_half<<<1024, 256>>>(w, (size_t)nmax * kmax, 99);
    CK(hipMalloc(&ws, ws_bytes));
    CK(hipDeviceSynchronize());

    hipblasLtHandle_t handle;
    CB(hipblasLtCreate(&handle));
    hipStream_t stream; CK(hipStreamCreate(&stream));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    Shape shapes[] = {{1152, 1152, 1152, "q|k|v"}, {768, 1152, 1152, "q|k into [T,1152]"}, {384, 1152, 1152, "v into [T,1152]"},
                      {1536, 1152, 1536, "ffn first"}, {768, 1152, 768, "n=768 dense out"}};
    for (const Shape& s : shapes) {
        hipblasLtMatmulDesc_t desc;
        CB(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
        CB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
        CB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
        hipblasLtMatrixLayout_t la, lb, lc;
        CB(hipblasLtMatrixLayoutCreate(&la, HIP_R_16F, s.k, s.n, s.k));
        CB(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16F, s.k, tokens, s.k));
        CB(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, s.n, tokens, s.ldo));
        float alpha = 1.f, beta = 0.f;

        hipblasLtMatmulPreference_t pref;
        CB(hipblasLtMatmulPreferenceCreate(&pref));
        CB(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)));
        hipblasLtMatmulHeuristicResult_t first[32]; int n_first = 0;
        CB(hipblasLtMatmulAlgoGetHeuristic(handle, desc, la, lb, lc, lc, pref, 32, first, &n_first));
        int first_index = n_first ? hipblaslt_ext::getIndexFromAlgo(first[0].algo) : -1;
        std::vector<int> wanted;
        for (int i = 0; i < n_first; ++i) wanted.push_back(hipblaslt_ext::getIndexFromAlgo(first[i].algo));

        std::vector<hipblasLtMatmulHeuristicResult_t> all;
        CB(hipblaslt_ext::getAllAlgos(handle, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, ta, tb, HIP_R_16F, HIP_R_16F, HIP_R_32F,
                                      HIP_R_32F, HIPBLAS_COMPUTE_32F, all));
        struct R { float ms; int index; std::string name; };
        std::vector<R> res;
        int tried = 0;
        for (auto& h : all) {
            size_t need = 0;
            if (hipblaslt_ext::matmulIsAlgoSupported(handle, desc, &alpha, la, lb, &beta, lc, lc, h.algo, need) != HIPBLAS_STATUS_SUCCESS
                || need > ws_bytes)
                continue;
            ++tried;
            auto run = [&](int i) {
                return hipblasLtMatmul(handle, desc, &alpha, w, la, x[i & 1], lb, &beta, out[i & 1], lc, out[i & 1], lc, &h.algo, ws,
                                       ws_bytes, stream);
            };
            if (run(0) != HIPBLAS_STATUS_SUCCESS) continue;
            CK(hipStreamSynchronize(stream));
            CK(hipEventRecord(e0, stream));
            run(1);
            CK(hipEventRecord(e1, stream));
            CK(hipStreamSynchronize(stream));
            float ms1; CK(hipEventElapsedTime(&ms1, e0, e1));
            int idx = hipblaslt_ext::getIndexFromAlgo(h.algo);
            float best_known = res.empty() ? 1e30f : std::min_element(res.begin(), res.end(), [](const R& a, const R& b) { return a.ms < b.ms; })->ms;
            float ms = ms1;
            if (ms1 < 1.3f * best_known || std::find(wanted.begin(), wanted.end(), idx) != wanted.end()) {   // worth a proper measurement
                const int reps = 6;
                CK(hipEventRecord(e0, stream));
                for (int i = 0; i < reps; ++i) run(i);
                CK(hipEventRecord(e1, stream));
                CK(hipStreamSynchronize(stream));
                CK(hipEventElapsedTime(&ms, e0, e1));
                ms /= reps;
            }
            res.push_back({ms, idx, hipblaslt_ext::getKernelNameFromAlgo(handle, h.algo)});
        }
        std::sort(res.begin(), res.end(), [](const R& a, const R& b) { return a.ms < b.ms; });
        double flop = 2.0 * tokens * s.n * s.k;
        printf("== %s: out[%ld, %d] (ld %d), k %d: %zu solutions, %d supported, library's first choice index %d\n", s.what, tokens, s.n,
               s.ldo, s.k, all.size(), tried, first_index);
        for (size_t i = 0; i < res.size(); ++i)
            if ((int)i < top || res[i].index == first_index)
                printf("  #%zu index %d  %.3f ms  %.0f TFLOP/s  %s%s\n", i, res[i].index, res[i].ms, flop / res[i].ms * 1e-9,
                       res[i].name.substr(0, 150).c_str(), res[i].index == first_index ? "   <-- first choice" : "");
        printf("  the library's own order (first %d):", n_first);
        for (int w : wanted)
            for (size_t i = 0; i < res.size(); ++i)
                if (res[i].index == w) printf(" #%zu/%.3f", i, res[i].ms);
        printf("\n");
        fflush(stdout);
        hipblasLtMatmulPreferenceDestroy(pref);
        hipblasLtMatrixLayoutDestroy(la); hipblasLtMatrixLayoutDestroy(lb); hipblasLtMatrixLayoutDestroy(lc);
        hipblasLtMatmulDescDestroy(desc);
    }
    hipblasLtDestroy(handle);
    return 0;
}
