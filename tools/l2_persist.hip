// Does an XCD's L2 keep clean lines across a kernel boundary?  (The next-weight prefetch of csrc/gemm.hip relies on it; docs/LABNOTES.md 4.10.)
// Launch A: the 32 blocks of XCD x (blockIdx % 8 == x) read slice x (1 MiB) of a buffer.  Launch B reads it again and times the read per block:
//   same  = slice x again (L2 hits if the lines survived the boundary), cross = slice (x + 1) % 8 (another XCD's L2 has them: Infinity Cache at best),
//   cold  = a slice of a second buffer nobody touched since a 512 MiB sweep (HBM).
// Stream-ordered launches and a captured graph.  build: hipcc -O2 --offload-arch=gfx950 tools/l2_persist.hip -o tools/l2_persist
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int SLICE = 1 << 20, NBLK = 256, PER_BLOCK = SLICE / 32;   // 32 KiB per block = 256 threads x 8 x 16 B

__global__ void touch(const uint4* buf, int shift, unsigned long long* t_out, uint4* sink) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const uint4* p = buf + ((size_t)((xcd + shift) & 7) * SLICE + (size_t)j * PER_BLOCK) / 16 + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    uint4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint4 v = p[i * 256];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (acc.x == 0x12345678u) sink[threadIdx.x] = acc;   // keep the loads
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (t_out && threadIdx.x == 0) t_out[blockIdx.x] = t1 - t0;
}
__global__ void sweep(uint4* big, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) big[i] = uint4{1, 2, 3, 4};
}

int main() {
    uint4 *buf, *buf2, *big, *sink;
    unsigned long long* t;
    const size_t bigb = 512ull << 20;
    CK(hipMalloc(&buf, 8 * SLICE)); CK(hipMalloc(&buf2, 8 * SLICE)); CK(hipMalloc(&big, bigb)); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&t, NBLK * 8));
    CK(hipMemset(buf, 1, 8 * SLICE)); CK(hipMemset(buf2, 1, 8 * SLICE));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    auto report = [&](const char* name) -> int {
        std::vector<unsigned long long> h(NBLK);
        CK(hipMemcpy(h.data(), t, NBLK * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        printf("  %-28s median %6.2f us  max %6.2f us per block (32 KiB)\n", name, h[NBLK / 2] / 100.0, h[NBLK - 1] / 100.0);   // s_memrealtime: 100 MHz
        return 0;
    };
    for (int mode = 0; mode < 2; ++mode) {
        printf("%s\n", mode == 0 ? "stream-ordered launches" : "captured graph");
        for (int variant = 0; variant < 3; ++variant) {
            const char* names[3] = {"same XCD's slice again", "another XCD's slice", "untouched buffer (HBM)"};
            hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, s, big, bigb / 16);   // evict everything (L2 and Infinity Cache)
            CK(hipStreamSynchronize(s));
            if (mode == 0) {
                hipLaunchKernelGGL(touch, dim3(NBLK), dim3(256), 0, s, buf, 0, (unsigned long long*)nullptr, sink);
                hipLaunchKernelGGL(touch, dim3(NBLK), dim3(256), 0, s, variant == 2 ? buf2 : buf, variant == 1 ? 1 : 0, t, sink);
                CK(hipStreamSynchronize(s));
            } else {
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
                hipLaunchKernelGGL(touch, dim3(NBLK), dim3(256), 0, s, buf, 0, (unsigned long long*)nullptr, sink);
                hipLaunchKernelGGL(touch, dim3(NBLK), dim3(256), 0, s, variant == 2 ? buf2 : buf, variant == 1 ? 1 : 0, t, sink);
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge, s));
                CK(hipStreamSynchronize(s));
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            }
            if (report(names[variant])) return 1;
        }
    }
    // How much may another launch stream through the L2 between the prefetch and the re-read before the lines are gone?  (The GEMM that issues the
    // prefetch streams ~2.5 MB per XCD of its own operands afterwards.)  touch slice x -> stream `mb` MiB per XCD of a third buffer -> re-read slice x.
    printf("re-read of the same XCD's 1 MiB slice after another launch streamed N MiB per XCD through the L2 (stream-ordered)\n");
    uint4* buf3;
    const size_t b3 = 16ull * (8 << 20);   // 16 regions of 8 MiB (1 MiB per XCD each); the loop below uses regions 0 .. 11
    CK(hipMalloc(&buf3, b3));
    CK(hipMemset(buf3, 1, b3));
    for (int half_mb = 0; half_mb <= 12; half_mb += (half_mb < 4 ? 1 : 2)) {
        hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, s, big, bigb / 16);
        CK(hipStreamSynchronize(s));
        hipLaunchKernelGGL(touch, dim3(NBLK), dim3(256), 0, s, buf, 0, (unsigned long long*)nullptr, sink);
        for (int k = 0; k < half_mb && k < 16; ++k)   // each launch reads one 8 MiB region = 1 MiB per XCD
            hipLaunchKernelGGL(touch, dim3(NBLK), dim3(256), 0, s, buf3 + (size_t)k * ((8 << 20) / 16), 0, (unsigned long long*)nullptr, sink);
        hipLaunchKernelGGL(touch, dim3(NBLK), dim3(256), 0, s, buf, 0, t, sink);
        CK(hipStreamSynchronize(s));
        char name[64];
        snprintf(name, sizeof(name), "%d MiB per XCD in between", half_mb);
        if (report(name)) return 1;
    }
    return 0;
}
