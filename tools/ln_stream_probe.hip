// Where does the LayerNorm launch lose bandwidth against a plain fp32 -> fp16 copy?  One 256-thread block per 4 KiB row (the shape of
// ln_row_block_kernel), pieces added one by one; M rows of 1024 floats from three rotating inputs (HBM).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/ln_stream_probe tools/ln_stream_probe.hip && /tmp/ln_stream_probe [M]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ size_t tiled_off(int r, int k, int K) {
    return ((size_t)(r >> 7) * (size_t)(K >> 6) + (size_t)(k >> 6)) * 8192 + (size_t)((r & 127) * 64) + (size_t)(((((k >> 3) & 7) ^ (r & 7)) << 3) + (k & 7));
}
__device__ __forceinline__ void store16(void* dst, u32x4 v, bool sc1) {
    if (sc1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
    else *(u32x4*)dst = v;
}
__device__ __forceinline__ void store_paired(f16* dst, f16x4 o, int lane, bool sc1) {
    union { f16x4 h; unsigned u[2]; } mine, other;
    mine.h = o;
    other.u[0] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine.u[0], 0xB1, 0xF, 0xF, true);
    other.u[1] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine.u[1], 0xB1, 0xF, 0xF, true);
    if (!(lane & 1)) store16(dst, u32x4{mine.u[0], mine.u[1], other.u[0], other.u[1]}, sc1);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// LAYOUT 0 linear 8-byte stores, 1 tiled 8-byte, 2 tiled paired 16-byte, 3 tiled paired 16-byte sc1; RED: block reduction + barrier; AFF: gamma / beta
template <int LAYOUT, bool RED, bool AFF>
__global__ __launch_bounds__(256) void row_block(const float* __restrict__ x, f16* __restrict__ out, int M, const float* __restrict__ ga, const float* __restrict__ be) {
    __shared__ float red[16];
    const int m = blockIdx.x, c = threadIdx.x * 4, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    f32x4 v = *(const f32x4*)(x + (size_t)m * 1024 + c);
    f32x4 a = f32x4{1.f, 1.f, 1.f, 1.f}, b = f32x4{0.f, 0.f, 0.f, 0.f};
    if (AFF) { a = *(const f32x4*)(ga + c); b = *(const f32x4*)(be + c); }
    float mean = 0.f, rstd = 1.f;
    if (RED) {
        const float s1 = wave_sum((v[0] + v[1]) + (v[2] + v[3]));
        const float s2 = wave_sum((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
        if (lane == 0) { red[wid] = s1; red[8 + wid] = s2; }
        __syncthreads();
        const float t1 = (red[0] + red[1]) + (red[2] + red[3]), t2 = (red[8] + red[9]) + (red[10] + red[11]);
        mean = t1 / 1024.f;
        rstd = 1.0f / sqrtf(fmaxf(t2 / 1024.f - mean * mean, 0.f) + 1e-6f);
    }
    f16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (f16)((v[e] - mean) * rstd * a[e] + b[e]);
    if (LAYOUT == 0) *(f16x4*)(out + (size_t)m * 1024 + c) = o;
    else if (LAYOUT == 1) *(f16x4*)(out + tiled_off(m, c, 1024)) = o;
    else store_paired(out + tiled_off(m, c, 1024), o, lane, LAYOUT == 3);
}

// R rows per thread, all loads first (the copy kernel's way); linear or tiled paired sc1 output; optional per-row reduction (R barriers' worth in one)
template <int R, int LAYOUT, bool RED>
__global__ __launch_bounds__(256) void rows_block(const float* __restrict__ x, f16* __restrict__ out, int M) {
    __shared__ float red[R][16];
    const int m0 = blockIdx.x * R, c = threadIdx.x * 4, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    f32x4 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = *(const f32x4*)(x + (size_t)(m0 + r) * 1024 + c);
    float mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { mean[r] = 0.f; rstd[r] = 1.f; }
    if (RED) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float s1 = wave_sum((v[r][0] + v[r][1]) + (v[r][2] + v[r][3]));
            const float s2 = wave_sum((v[r][0] * v[r][0] + v[r][1] * v[r][1]) + (v[r][2] * v[r][2] + v[r][3] * v[r][3]));
            if (lane == 0) { red[r][wid] = s1; red[r][8 + wid] = s2; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float t1 = (red[r][0] + red[r][1]) + (red[r][2] + red[r][3]), t2 = (red[r][8] + red[r][9]) + (red[r][10] + red[r][11]);
            mean[r] = t1 / 1024.f;
            rstd[r] = 1.0f / sqrtf(fmaxf(t2 / 1024.f - mean[r] * mean[r], 0.f) + 1e-6f);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)((v[r][e] - mean[r]) * rstd[r]);
        if (LAYOUT == 0) *(f16x4*)(out + (size_t)(m0 + r) * 1024 + c) = o;
        else store_paired(out + tiled_off(m0 + r, c, 1024), o, lane, LAYOUT == 3);
    }
}

// one WAVE per row, 4 float4 per lane (16 B x 64 lanes x 4 = the row), no LDS / barrier; 4 rows per 256-thread block
template <int LAYOUT, bool RED>
__global__ __launch_bounds__(256) void wave_row(const float* __restrict__ x, f16* __restrict__ out, int M) {
    const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *(const f32x4*)(x + (size_t)m * 1024 + i * 256 + lane * 4);
    float mean = 0.f, rstd = 1.f;
    if (RED) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { s1 += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); s2 += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]); }
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        mean = s1 / 1024.f;
        rstd = 1.0f / sqrtf(fmaxf(s2 / 1024.f - mean * mean, 0.f) + 1e-6f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (f16)((v[i][e] - mean) * rstd);
        const int c = i * 256 + lane * 4;
        if (LAYOUT == 0) *(f16x4*)(out + (size_t)m * 1024 + c) = o;
        else store_paired(out + tiled_off(m, c, 1024), o, lane, LAYOUT == 3);
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 46080;
    const size_t n = (size_t)M * 1024;
    float* x[3]; f16* out; float *ga, *be;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xFFFF) / 65536.f * 6.f - 2.5f;
    for (int i = 0; i < 3; ++i) { CK(hipMalloc(&x[i], n * 4)); CK(hipMemcpy(x[i], h.data(), n * 4, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&out, n * 2)); CK(hipMalloc(&ga, 4096)); CK(hipMalloc(&be, 4096));
    CK(hipMemset(ga, 0, 4096)); CK(hipMemset(be, 0, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 6; ++i) launch(i);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 60;
        for (int i = 0; i < reps; ++i) launch(i);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipGetLastError());
        const double us = ms * 1e3 / reps;
        printf("M=%6d %-72s %8.2f us  %5.2f TB/s\n", M, name, us, 6.0 * n / us / 1e6);
    };
#define RB(L, R_, A) timeit("block/row layout " #L " reduce " #R_ " affine " #A, [&](int i) { hipLaunchKernelGGL((row_block<L, R_, A>), dim3(M), dim3(256), 0, 0, x[i % 3], out, M, ga, be); })
    RB(0, false, false); RB(1, false, false); RB(2, false, false); RB(3, false, false);
    RB(0, true, false); RB(3, true, false); RB(3, true, true); RB(0, true, true);
#define RS(R, L, R_) timeit("block/" #R " rows layout " #L " reduce " #R_, [&](int i) { hipLaunchKernelGGL((rows_block<R, L, R_>), dim3(M / R), dim3(256), 0, 0, x[i % 3], out, M); })
    RS(2, 0, false); RS(4, 0, false); RS(8, 0, false); RS(4, 3, false); RS(4, 0, true); RS(4, 3, true); RS(8, 3, true);
#define WR(L, R_) timeit("wave/row layout " #L " reduce " #R_, [&](int i) { hipLaunchKernelGGL((wave_row<L, R_>), dim3(M / 4), dim3(256), 0, 0, x[i % 3], out, M); })
    WR(0, false); WR(3, false); WR(0, true); WR(3, true);
    return 0;
}
