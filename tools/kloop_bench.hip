// Compute side of a GEMM K-step in isolation (no global memory in the loop): LDS fragment reads (the tile-major swizzled image of
// csrc/gemm.hip) + v_mfma_f32_16x16x32_f16 + the workgroup barrier, one workgroup per CU on every CU.  What does a K-step of the
// 128 x 96 block tile cost in shader cycles when nothing is waited for but LDS and the matrix pipe?  Variants: waves per block
// and wave tile, software-pipeline depth (whole K-step / half K-step), reads blocked in front of the MFMAs or interleaved with
// them, barrier per K-step on / off, MFMAs off (read skeleton), reads off (MFMA only).
//   hipcc --offload-arch=gfx950 -O3 tools/kloop_bench.hip -o tools/kloop_bench && ./tools/kloop_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }

template <int NM, int ND, int RPM>
__device__ __forceinline__ void interleave() {
    constexpr int NG = (ND + RPM - 1) / RPM;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);
    }
    if constexpr (NM > NG) __builtin_amdgcn_sched_group_barrier(0x008, NM - NG, 0);
}

// MODE bits: 1 = barrier per K-step, 2 = reads, 4 = MFMAs, 8 = interleave reads with MFMAs (else reads first), 16 = two reads per MFMA slot
// XW extra waves do nothing but take part in the barriers (the loader waves of gemm_l_kernel when their fills are switched off)
template <int WN, int WM, int FI, int FJ, int MODE, int XW = 0>
__global__ __launch_bounds__(64 * (WN * WM + XW), 1) void kloop(int steps, float* out, unsigned long long* cyc, const char* src) {
    constexpr int NS = 4;
    constexpr int WPC = 2 * FI * WN, XPC = 2 * FJ * WM, STAGE = (WPC + XPC) * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = (tid >> 6) - XW;
    for (int i = tid; i < NS * STAGE / 4; i += blockDim.x) ((unsigned*)smem)[i] = 0x3c003c00u ^ ((unsigned)i * 2654435761u & 0x03ff03ffu);   // fp16 values near 1
    __syncthreads();
    if (w < 0) {
        __syncthreads();
        if constexpr ((MODE & 32) != 0 && XW > 0) {
            // real loader waves: every K-step each of the XW waves issues its share of a (WPC + XPC)-piece tile by LDS-DMA from an
            // L2-resident 2 MiB region into the ring (3 tiles in flight), like gemm_l_kernel's loaders
            constexpr int NP = WPC + XPC, G = (NP + XW - 1) / (XW > 0 ? XW : 1);
            const int lw = tid >> 6;
            size_t off = ((size_t)blockIdx.x * 37 * NP * 1024) % (2u << 20);
            auto stage = [&](int t) {
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    int q = lw * G + i;
                    q = q < NP ? q : NP - 1;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off + q * 1024 + lane * 16),
                                                     (__attribute__((address_space(3))) void*)(smem + (t % NS) * STAGE + q * 1024), 16, 0, 0);
                }
                off += NP * 1024;
                if (off + NP * 1024 > (2u << 20)) off = 0;
            };
            for (int t = 0; t < NS - 1; ++t) stage(t);
            for (int t = 0; t < steps; ++t) {
                __builtin_amdgcn_s_waitcnt(waitcnt_imm((NS - 2) * G, 15));
                asm volatile("s_barrier" ::: "memory");
                stage(t + NS - 1);
            }
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 15));
            return;
        }
        if (MODE & 1) for (int t = 0; t < steps; ++t) asm volatile("s_barrier" ::: "memory");
        return;
    }
    const int wn = w % WN, wm = w / WN, li = lane & 15, g = lane >> 4;
    int woff[2], xoff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int ch = ((4 * s + g) ^ (li & 7)) << 4;
        woff[s] = (16 * FI * wn + li) * 128 + ch;
        xoff[s] = WPC * 1024 + (16 * FJ * wm + li) * 128 + ch;
    }
    f32x4 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 wA[2][FI], xA[2][FJ], wB[2][FI], xB[2][FJ];
    auto rd = [&](int t, f16x8 (&wf)[2][FI], f16x8 (&xf)[2][FJ]) {
        const char* b = smem + (t % NS) * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < FI; ++i) wf[s][i] = *(const f16x8*)(b + woff[s] + i * 16 * 128);
#pragma unroll
            for (int j = 0; j < FJ; ++j) xf[s][j] = *(const f16x8*)(b + xoff[s] + j * 16 * 128);
        }
    };
    auto mm = [&](const f16x8 (&wf)[2][FI], const f16x8 (&xf)[2][FJ]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[s][i], xf[s][j], acc[i][j], 0, 0, 0);
    };
    rd(0, wB, xB);
    rd(1, wA, xA);
    auto step = [&](int t, f16x8 (&wr)[2][FI], f16x8 (&xr)[2][FJ], const f16x8 (&wm_)[2][FI], const f16x8 (&xm_)[2][FJ]) {
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
        if (MODE & 1) asm volatile("s_barrier" ::: "memory");
        if (MODE & 2) rd(t, wr, xr);
        if (!(MODE & 8)) __builtin_amdgcn_sched_barrier(0);
        if (MODE & 4) mm(wm_, xm_);
        if ((MODE & 8) && (MODE & 2) && (MODE & 4)) interleave<2 * FI * FJ, 2 * (FI + FJ), (MODE & 16) ? 2 : 1>();
        __builtin_amdgcn_sched_barrier(0);
    };
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < steps; t += 2) {
        step(t, wA, xA, wB, xB);
        step(t + 1, wB, xB, wA, xA);
    }
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) sum += acc[i][j];
    if (!(MODE & 4)) {   // keep the reads alive
        union { f16x8 h; f32x4 f; } u;
        u.h = wA[0][0] + wB[1][FI - 1] + xA[0][0] + xB[1][FJ - 1];
        sum += u.f;
    }
    out[(size_t)blockIdx.x * blockDim.x + tid] = sum[0] + sum[1] + sum[2] + sum[3];
    if (w == 0 && lane == 0) { cyc[blockIdx.x] = c1 - c0; cyc[gridDim.x + blockIdx.x] = r1 - r0; }
}

static char* g_src = nullptr;
template <int WN, int WM, int FI, int FJ, int MODE, int XW = 0>
static void run(const char* what) {
    const int steps = 4000, cus = 256;
    constexpr int LDS = 4 * (2 * FI * WN + 2 * FJ * WM) * 1024;
    CK(hipFuncSetAttribute((const void*)kloop<WN, WM, FI, FJ, MODE, XW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    float* out;
    unsigned long long* cyc;
    CK(hipMalloc((void**)&out, (size_t)cus * 1024 * 4));
    CK(hipMalloc((void**)&cyc, cus * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((kloop<WN, WM, FI, FJ, MODE, XW>), dim3(cus), dim3(64 * (WN * WM + XW)), LDS, 0, steps, out, cyc, (const char*)g_src);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h(2 * cus);
    CK(hipMemcpy(h.data(), cyc, cus * 16, hipMemcpyDeviceToHost));
    double c = 0, rt = 0;
    for (int i = 0; i < cus; ++i) { c += (double)h[i]; rt += (double)h[cus + i]; }
    c /= cus;
    rt /= cus;                      // 10 ns ticks
    const double mfma_cyc = 2.0 * FI * FJ * WN * WM / 4 * 16;   // per SIMD per K-step at 16 cycles per MFMA
    printf("%-58s %d waves %dx%d tiles: %7.1f cycles = %5.3f us per K-step (MFMA alone %4.0f cycles, LDS reads alone %4.0f), %.2f GHz\n", what, WN * WM + XW, 16 * FI,
           16 * FJ, c / steps, rt * 0.01 / steps, (MODE & 4) ? mfma_cyc : 0.0, (MODE & 2) ? 2.0 * (FI + FJ) * WN * WM * 4 : 0.0, c / (rt * 10.0));
    CK(hipFree(out));
    CK(hipFree(cyc));
}

int main() {
    CK(hipMalloc((void**)&g_src, (2u << 20) + (1u << 20)));
    CK(hipMemset(g_src, 0x3c, (2u << 20) + (1u << 20)));
    // 128 x 96 block tile
    run<4, 2, 2, 3, 1 | 2 | 4>("8 waves of 32x48, barrier, reads first");
    run<4, 2, 2, 3, 1 | 2 | 4 | 8>("8 waves of 32x48, barrier, interleaved 1/MFMA");
    run<4, 2, 2, 3, 1 | 2 | 4 | 8 | 16>("8 waves of 32x48, barrier, interleaved 2/MFMA");
    run<4, 2, 2, 3, 1 | 2 | 4 | 8, 4>("8 waves of 32x48 + 4 barrier-only waves, interleaved 1/MFMA");
    run<4, 2, 2, 3, 1 | 2 | 4 | 8 | 32, 4>("8 waves of 32x48 + 4 LOADER waves (28 KiB / K-step by LDS-DMA)");
    run<4, 2, 2, 3, 1 | 2 | 32, 4>("8 waves of 32x48 reads only + 4 LOADER waves");
    run<4, 2, 2, 3, 1 | 32, 4>("8 waves barrier only + 4 LOADER waves (fill stream alone)");
    run<4, 2, 2, 3, 1 | 4 | 32, 4>("8 waves of 32x48 MFMA only + 4 LOADER waves");
    run<2, 2, 4, 3, 1 | 2 | 4 | 8 | 32, 4>("4 waves of 64x48 + 4 LOADER waves");
    run<4, 2, 2, 3, 2 | 4 | 8 | 16>("8 waves of 32x48, NO barrier, interleaved 2/MFMA");
    run<4, 2, 2, 3, 1 | 4>("8 waves of 32x48, barrier, MFMA only");
    run<4, 2, 2, 3, 4>("8 waves of 32x48, MFMA only, no barrier");
    run<4, 2, 2, 3, 1 | 2>("8 waves of 32x48, barrier, reads only");
    run<4, 2, 2, 3, 1>("8 waves of 32x48, barrier only");
    run<2, 2, 4, 3, 1 | 2 | 4>("4 waves of 64x48, barrier, reads first");
    run<2, 2, 4, 3, 1 | 2 | 4 | 8>("4 waves of 64x48, barrier, interleaved 1/MFMA");
    run<2, 2, 4, 3, 1 | 2 | 4 | 8 | 16>("4 waves of 64x48, barrier, interleaved 2/MFMA");
    run<2, 2, 4, 3, 2 | 4 | 8 | 16>("4 waves of 64x48, NO barrier, interleaved 2/MFMA");
    run<2, 2, 4, 3, 1 | 4>("4 waves of 64x48, barrier, MFMA only");
    run<2, 2, 4, 3, 4>("4 waves of 64x48, MFMA only, no barrier");
    run<2, 2, 4, 3, 1 | 2>("4 waves of 64x48, barrier, reads only");
    run<2, 3, 4, 2, 1 | 2 | 4 | 8 | 16>("6 waves of 64x32 (shape 9), barrier, interleaved 2/MFMA");
    return 0;
}
