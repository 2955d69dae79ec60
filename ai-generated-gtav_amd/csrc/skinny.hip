// Exact-fp32 "skinny" GEMM for the conditioning path (timestep MLP, action Linear, the adaLN
// mega-projection):  Y[m][n] = act( sum_k X[m][k] * W[n][k] + bias[n] ),  M = B*T rows (tens), N up to
// 198 656, K ~ 1 K.  The fp16 experiment in DESIGN.md shows this path alone costs 9.5e-4 rel-L2 if
// its operands are rounded to fp16, so it runs on v_mfma_f32_16x16x4_f32 (bit-for-bit an fp32 fma
// chain) with fp32 weights.  It is weight-streaming (HBM) bound: each wave streams 16 rows of W
// straight into registers (never reused), X (16 rows) sits in LDS.
#include "ops.h"

namespace gtav {

namespace {

// One block = 4 waves = 64 output features x one 16-row slab of X.
template <int ACT>
__global__ __launch_bounds__(256) void skinny_f32_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                        int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [16][K + 8]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 16;
    const int n0 = blockIdx.x * 64 + w * 16;
    const int ldk = K + 8;

    // stage the X slab (rows beyond M are zero)
    const int kq = K >> 2;
    for (int idx = tid; idx < 16 * kq; idx += 256) {
        const int r = idx / kq, c = idx - r * kq;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m0 + r < M) v = *(const f32x4*)(X + (size_t)(m0 + r) * ldx + 4 * c);
        *(f32x4*)(xs + r * ldk + 4 * c) = v;
    }
    __syncthreads();
    if (n0 >= N) return;

    int nrow = n0 + li;
    nrow = nrow < N ? nrow : N - 1;
    const float* wp = W + (size_t)nrow * K + 4 * g;
    const float* xp = xs + li * ldk + 4 * g;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    // K % 32 == 0; each step covers 32 k: lane group g holds k = kb + 16 i + 4 g + e
#pragma unroll 4
    for (int kb = 0; kb < K; kb += 32) {
        const f32x4 w0 = *(const f32x4*)(wp + kb);
        const f32x4 w1 = *(const f32x4*)(wp + kb + 16);
        const f32x4 x0 = *(const f32x4*)(xp + kb);
        const f32x4 x1 = *(const f32x4*)(xp + kb + 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[e], x0[e], acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[e], x1[e], acc, 0, 0, 0);
    }
    // D[row = feature 4g + r][col = X row li]
    const int m = m0 + li, n = n0 + 4 * g;
    if (m < M && n < N) {
        f32x4 v = acc;
        if (bias) v = v + *(const f32x4*)(bias + n);
        if (ACT == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f + expf(-v[e]));
        }
        *(f32x4*)(Y + (size_t)m * ldy + n) = v;
    }
}

}  // namespace

int skinny_init() {
    static unsigned long long done_devs = 0;   // the attribute is per device (see attention.hip)
    int devid = 0;
    GTAV_CHECK_HIP(hipGetDevice(&devid));
    if (done_devs >> (devid & 63) & 1) return 0;
    GTAV_CHECK_HIP(hipFuncSetAttribute((const void*)skinny_f32_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    GTAV_CHECK_HIP(hipFuncSetAttribute((const void*)skinny_f32_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done_devs |= 1ull << (devid & 63);
    return 0;
}

int launch_skinny_f32(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int M, int N,
                      int K, int act_silu, hipStream_t stream) {
    GTAV_REQUIRE(M > 0 && N > 0 && N % 4 == 0, "skinny: bad M=%d N=%d", M, N);
    GTAV_REQUIRE(K > 0 && K % 32 == 0 && K <= 2048, "skinny: K=%d must be a multiple of 32 and <= 2048", K);
    GTAV_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ldx >= K && ldy >= N, "skinny: bad leading dims");
    dim3 grid(cdiv(N, 64), cdiv(M, 16)), block(256);
    const size_t lds = (size_t)16 * (K + 8) * sizeof(float);
    if (act_silu)
        hipLaunchKernelGGL(skinny_f32_kernel<1>, grid, block, lds, stream, X, ldx, W, bias, Y, ldy, M, N, K);
    else
        hipLaunchKernelGGL(skinny_f32_kernel<0>, grid, block, lds, stream, X, ldx, W, bias, Y, ldy, M, N, K);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace gtav
