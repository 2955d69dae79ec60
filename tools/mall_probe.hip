// How long does a line that one launch READ stay in the Infinity Cache (MALL) on THIS GPU?  (Box survey, round 5: the next-weight prefetch pays on some
// MI355X GPUs and not on others, and an XCD shift of the prefetched slices does not change its gain where it pays — so the gain is not the XCD-local L2.)
// touch A (8 MiB, read by XCD x's blocks) -> read N MiB of another buffer -> re-read A from the blocks of XCD x + 1 (never an L2 hit) and time each block's 32 KiB.
// build + run: hipcc -O2 --offload-arch=gfx950 tools/mall_probe.hip -o /tmp/mall_probe && /tmp/mall_probe
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
    if (acc.x == 0x12345678u) sink[threadIdx.x] = acc;
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (t_out && threadIdx.x == 0) t_out[blockIdx.x] = t1 - t0;
}
__global__ void stream_read(const uint4* big, size_t n, uint4* sink) {
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = big[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (acc.x == 0x12345678u) sink[threadIdx.x] = acc;
}
__global__ void sweep(uint4* big, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) big[i] = uint4{1, 2, 3, 4};
}

int main() {
    uint4 *buf, *fill, *big, *sink;
    unsigned long long* t;
    const size_t bigb = 1024ull << 20, fillb = 1024ull << 20;
    CK(hipMalloc(&buf, 8 * SLICE)); CK(hipMalloc(&fill, fillb)); CK(hipMalloc(&big, bigb)); CK(hipMalloc(&sink, 4096)); CK(hipMalloc(&t, NBLK * 8));
    CK(hipMemset(buf, 1, 8 * SLICE)); CK(hipMemset(fill, 1, fillb));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    auto report = [&](const char* name) -> int {
        std::vector<unsigned long long> h(NBLK);
        CK(hipMemcpy(h.data(), t, NBLK * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        printf("  %-44s median %6.2f us  p90 %6.2f  max %6.2f us per block (32 KiB)\n", name, h[NBLK / 2] / 100.0, h[NBLK * 9 / 10] / 100.0, h[NBLK - 1] / 100.0);
        return 0;
    };
    const int mbs[] = {0, 8, 32, 64, 128, 192, 256, 384, 512, 768};
    for (int rep = 0; rep < 2; ++rep)
    for (int mb : mbs) {
        hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, s, big, bigb / 16);   // evict everything (L2 and Infinity Cache)
        CK(hipStreamSynchronize(s));
        hipLaunchKernelGGL(touch, dim3(NBLK), dim3(256), 0, s, buf, 0, (unsigned long long*)nullptr, sink);
        if (mb) hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, s, fill, ((size_t)mb << 20) / 16, sink);
        hipLaunchKernelGGL(touch, dim3(NBLK), dim3(256), 0, s, buf, 1, t, sink);
        CK(hipStreamSynchronize(s));
        char name[96];
        snprintf(name, sizeof(name), "read %4d MiB in between, re-read from XCD+1", mb);
        if (report(name)) return 1;
    }
    // the sweep itself: is a line that was just WRITTEN served from the Infinity Cache?
    hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, s, big, bigb / 16);
    hipLaunchKernelGGL(sweep, dim3(256), dim3(256), 0, s, buf, (size_t)8 * SLICE / 16);
    hipLaunchKernelGGL(touch, dim3(NBLK), dim3(256), 0, s, buf, 1, t, sink);
    CK(hipStreamSynchronize(s));
    if (report("written by the launch before, read from XCD+1")) return 1;
    return 0;
}
