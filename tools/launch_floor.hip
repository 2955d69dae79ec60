// Measures what a dependent step costs on this GPU (docs/LABNOTES.md 9): (a) a captured hipGraph of N dependent kernels
// (256 blocks x 256 threads, each block stores one word) against (b) ONE persistent launch of 256 blocks that meets at N grid-wide
// barriers (one device-scope counter per barrier: every block's thread 0 adds 1 and polls with sc1 loads until all 256 arrived) and
// (c) the same barrier between TWO byte-moving phases (each block writes 4 KiB with write-through stores, the next phase reads another
// block's 4 KiB with sc1 loads: what a LayerNorm -> GEMM hand-off would need).  Build: hipcc -O2 --offload-arch=gfx950 tools/launch_floor.hip -o tools/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void tiny_kernel(unsigned* out, unsigned v) {
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);   // device-scope release: this block's stores are visible before the count
        int spins = 0;
        while (__builtin_nontemporal_load(counter) < nblocks) {   // coherent load each time
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) break;   // exit condition every wave reaches even if a block never arrives
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void persistent_barriers(unsigned* counters, int nbar, unsigned* out) {
    for (int i = 0; i < nbar; ++i) grid_barrier(counters + 32 * i, gridDim.x);
    if (threadIdx.x == 0) out[blockIdx.x] = nbar;
}

__global__ __launch_bounds__(256) void persistent_handoff(unsigned* counters, int nbar, u32x4* buf, unsigned* out) {
    // phase i: write this block's 4 KiB slot of buffer (i & 1), barrier, read the neighbour's slot
    unsigned acc = 0;
    for (int i = 0; i < nbar; ++i) {
        u32x4* mine = buf + ((size_t)(i & 1) * gridDim.x + blockIdx.x) * 256;
        const u32x4 v = u32x4{(unsigned)i, blockIdx.x, threadIdx.x, 1u};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(mine + threadIdx.x), "v"(v) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        grid_barrier(counters + 32 * i, gridDim.x);
        const u32x4* theirs = buf + ((size_t)(i & 1) * gridDim.x + (blockIdx.x + 97) % gridDim.x) * 256 + threadIdx.x;
        u32x4 r;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(theirs) : "memory");
        acc += r[0] + r[1];
    }
    if (acc == 0xFFFFFFFFu) out[blockIdx.x] = acc;
    if (threadIdx.x == 0) out[blockIdx.x] = nbar;
}

int main() {
    const int N = 200, NB = 256;
    unsigned *out, *counters;
    u32x4* buf;
    CK(hipMalloc(&out, NB * sizeof(unsigned)));
    CK(hipMalloc(&counters, (size_t)N * 32 * sizeof(unsigned)));
    CK(hipMalloc(&buf, (size_t)2 * NB * 256 * sizeof(u32x4)));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // (a) graph of N dependent kernels
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(NB), dim3(256), 0, s, out, (unsigned)i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("(a) graph of %d dependent tiny kernels (256 x 256): %.2f us per kernel\n", N, ms * 1e3 / N);
    }
    // (b) persistent kernel with N grid barriers
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(counters, 0, (size_t)N * 32 * sizeof(unsigned), s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(persistent_barriers, dim3(NB), dim3(256), 0, s, counters, N, out);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned> h(NB);
        CK(hipMemcpy(h.data(), out, NB * sizeof(unsigned), hipMemcpyDeviceToHost));
        printf("(b) one launch, %d grid barriers of 256 blocks: %.2f us per barrier (blocks done: %u)\n", N, ms * 1e3 / N, h[0]);
    }
    // (c) barrier + 4 KiB write-through / sc1 read hand-off per phase
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(counters, 0, (size_t)N * 32 * sizeof(unsigned), s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(persistent_handoff, dim3(NB), dim3(256), 0, s, counters, N, buf, out);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("(c) one launch, %d phases of [4 KiB sc1 store per block | grid barrier | 4 KiB sc1 load of another block's slot]: %.2f us per phase\n", N, ms * 1e3 / N);
    }
    return 0;
}
