// What does the memory system allow for the LayerNorm's traffic pattern at M = 5760 (B = 8)?  Per row of 1024 features: read the fp32 residual row and one
// fp32 slab row (4 KiB each), write the residual back (4 KiB) and the fp16 operand (2 KiB).  The product kernel (csrc/elementwise.hip ln_row_block_kernel)
// takes 17.5-18.4 us per launch = 4.6 TB/s; is that the fabric, or the kernel's structure (one short-lived block per row)?
//   a: one block of 256 threads per row, no arithmetic (pure traffic, same stores: write-through sc1)
//   b: a + a block-wide reduction between load and store (the statistics' barrier)
//   c: persistent blocks (8 per CU), each walking rows with the NEXT row's loads in flight while the current one is reduced and stored
// build: hipcc -O2 --offload-arch=gfx950 tools/ln_stream.hip -o tools/ln_stream
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int D = 1024;

__device__ __forceinline__ void st_sc1_16(void* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_sc1_8(void* p, f16x4 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

template <int MODE, bool NT = true>
__global__ __launch_bounds__(256) void ln_like(float* R, const float* S, _Float16* A, int M, const float* T) {
    __shared__ float red[8];
    const int c = threadIdx.x * 4;
    if (MODE < 2 || MODE >= 3) {
        const int m = blockIdx.x;
        f32x4 r = NT ? __builtin_nontemporal_load((const f32x4*)(R + (size_t)m * D + c)) : *(const f32x4*)(R + (size_t)m * D + c);
        f32x4 s = NT ? __builtin_nontemporal_load((const f32x4*)(S + (size_t)m * D + c)) : *(const f32x4*)(S + (size_t)m * D + c);
        f32x4 v = r + s;
        if (MODE >= 3) {   // + the product kernel's per-frame / per-layer vectors (shift, scale, gate, bias: L2 hits) and, MODE 4, its second read of the slab
            const float* tr = T + (size_t)(m / 144) * 4096;
            const f32x4 a = *(const f32x4*)(tr + c), b = *(const f32x4*)(tr + 1024 + c), g = *(const f32x4*)(tr + 2048 + c), bi = *(const f32x4*)(T + 3072 + c);
            v = (v + bi) * g + a * b;
            if (MODE == 4) v = v + *(const volatile f32x4*)(S + (size_t)m * D + c);
        }
        float k = 1.f;
        if (MODE >= 1) {
            float p = (v[0] + v[1]) + (v[2] + v[3]);
            for (int o = 32; o; o >>= 1) p += __shfl_xor(p, o, 64);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = p;
            __syncthreads();
            k = (red[0] + red[1]) + (red[2] + red[3]);
        }
        st_sc1_16(R + (size_t)m * D + c, v);
        st_sc1_8(A + (size_t)m * D + c, f16x4{(_Float16)(v[0] * k), (_Float16)(v[1] * k), (_Float16)(v[2] * k), (_Float16)(v[3] * k)});
    } else {
        int m = blockIdx.x;
        if (m >= M) return;
        f32x4 r = __builtin_nontemporal_load((const f32x4*)(R + (size_t)m * D + c));
        f32x4 s = __builtin_nontemporal_load((const f32x4*)(S + (size_t)m * D + c));
        for (; m < M; m += gridDim.x) {
            const int mn = m + gridDim.x;
            f32x4 rn = r, sn = s;
            if (mn < M) {
                rn = __builtin_nontemporal_load((const f32x4*)(R + (size_t)mn * D + c));
                sn = __builtin_nontemporal_load((const f32x4*)(S + (size_t)mn * D + c));
            }
            f32x4 v = r + s;
            float p = (v[0] + v[1]) + (v[2] + v[3]);
            for (int o = 32; o; o >>= 1) p += __shfl_xor(p, o, 64);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = p;
            __syncthreads();
            const float k = (red[0] + red[1]) + (red[2] + red[3]);
            st_sc1_16(R + (size_t)m * D + c, v);
            st_sc1_8(A + (size_t)m * D + c, f16x4{(_Float16)(v[0] * k), (_Float16)(v[1] * k), (_Float16)(v[2] * k), (_Float16)(v[3] * k)});
            r = rn; s = sn;
        }
    }
}

int main() {
    const int M = 5760;
    float *R, *S;
    _Float16* A;
    CK(hipMalloc(&R, (size_t)M * D * 4)); CK(hipMalloc(&S, (size_t)M * D * 4)); CK(hipMalloc(&A, (size_t)M * D * 2));
    CK(hipMemset(R, 0, (size_t)M * D * 4)); CK(hipMemset(S, 0, (size_t)M * D * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* T;
    CK(hipMalloc(&T, 64 * 4096 * 4)); CK(hipMemset(T, 0, 64 * 4096 * 4));
    const char* names[8] = {"a: block per row, traffic only", "b: block per row + reduction", "c: persistent, next row in flight", "d: b + four table vectors per thread", "e: d + the slab read twice", "f: b with plain (temporal) loads", "g: d with plain loads", "h: e with plain loads"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 8; ++mode) {
            const int n = 200;
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(ln_like<0>, dim3(M), dim3(256), 0, 0, R, S, A, M, T);
                else if (mode == 1) hipLaunchKernelGGL(ln_like<1>, dim3(M), dim3(256), 0, 0, R, S, A, M, T);
                else if (mode == 2) hipLaunchKernelGGL(ln_like<2>, dim3(2048), dim3(256), 0, 0, R, S, A, M, T);
                else if (mode == 3) hipLaunchKernelGGL(ln_like<3>, dim3(M), dim3(256), 0, 0, R, S, A, M, T);
                else if (mode == 4) hipLaunchKernelGGL(ln_like<4>, dim3(M), dim3(256), 0, 0, R, S, A, M, T);
                else if (mode == 5) hipLaunchKernelGGL((ln_like<1, false>), dim3(M), dim3(256), 0, 0, R, S, A, M, T);
                else if (mode == 6) hipLaunchKernelGGL((ln_like<3, false>), dim3(M), dim3(256), 0, 0, R, S, A, M, T);
                else hipLaunchKernelGGL((ln_like<4, false>), dim3(M), dim3(256), 0, 0, R, S, A, M, T);
            };
            for (int i = 0; i < 10; ++i) launch();
            CK(hipEventRecord(e0));
            for (int i = 0; i < n; ++i) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / n, bytes = (double)M * D * (4 + 4 + 4 + 2);
            printf("%-38s %7.2f us per launch  %.2f TB/s\n", names[mode], us, bytes / us * 1e-6);
        }
    return 0;
}
