// Per-CU operand ingest rates on gfx950: L2-resident bytes into a CU by (1) direct-to-LDS DMA (global_load_lds_dwordx4, the GEMM's
// fill path), (2) plain 16-byte loads to VGPRs, (3) both at once.  Question behind it: the GEMM main loops
// sit at the ~65-70 GB/s per-CU LDS-DMA ceiling; if the register path has its own, higher ceiling and the two add, one MFMA operand
// (the weights, whose fragments need no sharing between waves that own distinct features) can bypass LDS.
//   hipcc --offload-arch=gfx950 -O3 tools/ingest_bw.hip -o tools/ingest_bw && ./tools/ingest_bw
// Every wave-instruction moves 1 KiB of contiguous bytes (the shape of a fragment-major weight stream / a tile-major LDS piece).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// NL: LDS-DMA pieces per wave per step, NV: register loads (16 B per lane = 1 KiB per wave) per wave per step; ring depth 3 steps
template <int NL, int NV, int NWAVE>
__global__ __launch_bounds__(64 * NWAVE) void ingest(const char* __restrict__ src, size_t span, int steps, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = 3;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // every block walks the same L2-resident region from its own starting point (blocks of one XCD share it, like a W panel)
    const size_t step_bytes = (size_t)(NL + NV) * NWAVE * 1024;
    size_t off = ((size_t)blockIdx.x * 37 * step_bytes) % span;
    const char* base = src + (size_t)w * (NL + NV) * 1024 + lane * 16;
    u32x4 r[D][NV > 0 ? NV : 1];
    u32x4 acc = {0, 0, 0, 0};
    auto issue = [&](int s, int slot) {
        const char* p = base + off;
#pragma unroll
        for (int i = 0; i < NL; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(p + i * 1024), (lptr_t)(smem + ((slot * NWAVE + w) * NL + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i)   // asm: hipcc's own waits turn into vmcnt(0) next to LDS-DMA; completion is counted by hand below
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[slot][i]) : "v"(p + (NL + i) * 1024) : "memory");
        off += step_bytes;
        if (off + step_bytes > span) off = 0;
    };
#pragma unroll
    for (int s = 0; s < D - 1; ++s) issue(s, s);
    for (int s = 0; s < steps; s += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            issue(s + u + D - 1, (u + D - 1) % D);
            // consume step s + u: registers via a data dependence (the compiler's counted vmcnt), LDS pieces via a counted wait
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * (NL + NV)) : "memory");
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                asm volatile("" : "+v"(r[u][i]));   // the registers are valid from here on (after the counted wait)
                acc ^= r[u][i];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
    if (NL > 0 && ((const unsigned*)smem)[threadIdx.x] == 0x12345678u) sink[1] = 1;
}

template <int NL, int NV, int NWAVE>
static void run(const char* src, size_t span, unsigned* sink, int blocks_per_cu, const char* what) {
    const int steps = 1500, cus = 256;
    const int lds = NL > 0 ? 3 * NWAVE * NL * 1024 : 64;
    CK(hipFuncSetAttribute((const void*)ingest<NL, NV, NWAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const dim3 grid(cus * blocks_per_cu);
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((ingest<NL, NV, NWAVE>), grid, dim3(64 * NWAVE), lds, 0, src, span, steps, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes_cu = (double)steps * (NL + NV) * NWAVE * 1024 * blocks_per_cu;
    printf("%-44s waves/blk %d blk/CU %d  LDS-DMA %2d KiB + VGPR %2d KiB per step: %7.1f GB/s per CU (%5.2f TB/s chip), %.3f ms\n", what, NWAVE,
           blocks_per_cu, NL * NWAVE, NV * NWAVE, bytes_cu / (ms * 1e-3) / 1e9, bytes_cu * cus / (ms * 1e-3) / 1e12, ms);
}

int main() {
    const size_t span = 2u << 20;   // 2 MiB: resident in every XCD's 4 MiB L2
    char* src;
    unsigned* sink;
    CK(hipMalloc((void**)&src, span + (1 << 20)));
    CK(hipMalloc((void**)&sink, 64));
    std::vector<unsigned> h((span + (1 << 20)) / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u);
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(sink, 0, 64));
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run<4, 0, 4>(src, span, sink, bpc, "LDS-DMA only");
        run<8, 0, 4>(src, span, sink, bpc, "LDS-DMA only, 8 pieces per wave");
        run<0, 4, 4>(src, span, sink, bpc, "VGPR only");
        run<0, 8, 4>(src, span, sink, bpc, "VGPR only, 8 loads per wave");
        run<4, 4, 4>(src, span, sink, bpc, "both 1:1");
        run<3, 4, 4>(src, span, sink, bpc, "both 3:4 (X 96 rows via LDS, W 128 via VGPR)");
        run<2, 4, 4>(src, span, sink, bpc, "both 1:2");
        run<2, 6, 4>(src, span, sink, bpc, "both 1:3");
        run<2, 0, 8>(src, span, sink, bpc, "LDS-DMA only, 8 waves");
        run<0, 2, 8>(src, span, sink, bpc, "VGPR only, 8 waves");
        run<0, 4, 8>(src, span, sink, bpc, "VGPR only, 8 waves, 4 loads");
        run<2, 2, 8>(src, span, sink, bpc, "both 1:1, 8 waves");
        run<1, 2, 8>(src, span, sink, bpc, "both 1:2, 8 waves");
    }
    // HBM / Infinity-Cache-resident source for comparison (64 MiB region: beyond the L2s)
    {
        const size_t big = 64u << 20;
        char* src2;
        CK(hipMalloc((void**)&src2, big + (1 << 20)));
        CK(hipMemset(src2, 1, big + (1 << 20)));
        run<4, 0, 4>(src2, big, sink, 1, "LDS-DMA only, 64 MiB region");
        run<0, 4, 4>(src2, big, sink, 1, "VGPR only, 64 MiB region");
        run<2, 4, 4>(src2, big, sink, 1, "both 1:2, 64 MiB region");
    }
    return 0;
}
