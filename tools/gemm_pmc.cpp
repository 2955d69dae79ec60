// Torch-free driver for rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE) on the dominant kernel: runs the fc1-shaped
// GEMM (N = 4096, K = 1024, GELU epilogue) through the C-ABI at M = 720 and M = 5760 with rotating weight buffers.
//   hipcc -O2 tools/gemm_pmc.cpp -Iinclude -L ai-generated-gtav_amd -lgtav_amd -Wl,-rpath,'$ORIGIN/../ai-generated-gtav_amd' -o tools/gemm_pmc
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./tools/gemm_pmc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gtav_amd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int N = 4096, K = 1024, copies = 8, iters = argc > 1 ? atoi(argv[1]) : 16;
    const int Ms[2] = {720, 5760};
    std::vector<void*> w(copies);
    std::vector<unsigned short> host((size_t)N * K);
    for (size_t i = 0; i < host.size(); ++i) host[i] = 0x2000 + (unsigned short)((i * 2654435761u) >> 20 & 0x3ff);  // small fp16 values
    for (int c = 0; c < copies; ++c) {
        CK(hipMalloc(&w[c], (size_t)N * K * 2));
        CK(hipMemcpy(w[c], host.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
    }
    float* bias;
    CK(hipMalloc((void**)&bias, N * 4));
    CK(hipMemset(bias, 0, N * 4));
    for (int mi = 0; mi < 2; ++mi) {
        const int M = Ms[mi], Mp = (M + 127) / 128 * 128;
        void *x, *out;
        CK(hipMalloc(&x, (size_t)Mp * K * 2));
        CK(hipMemcpy(x, host.data(), (size_t)Mp * K * 2 < host.size() * 2 ? (size_t)Mp * K * 2 : host.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&out, (size_t)Mp * N * 2));
        for (int it = 0; it < iters; ++it) {
            if (gtav_op_gemm_f16(x, K, w[it % copies], bias, out, N, M, N, K, 2, nullptr, 0, 1, nullptr)) {
                fprintf(stderr, "gemm failed: %s\n", gtav_last_error());
                return 1;
            }
        }
        CK(hipDeviceSynchronize());
        printf("M=%d: %d launches of fc1 GEMM (N=%d K=%d), algorithmic bytes per launch: W %zu + X %zu + out %zu\n", M, iters, N, K,
               (size_t)N * K * 2, (size_t)M * K * 2, (size_t)M * N * 2);
        CK(hipFree(x));
        CK(hipFree(out));
    }
    return 0;
}
