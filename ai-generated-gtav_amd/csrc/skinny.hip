// Exact-fp32 "skinny" GEMM for the conditioning path (timestep MLP, action Linear, the adaLN
// mega-projection):  Y[m][n] = act( sum_k X[m][k] * W[n][k] + bias[n] ),  M = B*T rows (tens), N up to
// 198 656, K ~ 1 K.  The fp16 experiment in DESIGN.md shows this path alone costs 9.5e-4 rel-L2 if
// its operands are rounded to fp16, so it runs on v_mfma_f32_16x16x4_f32 (bit-for-bit an fp32 fma
// chain) with fp32 weights.  It is weight-streaming (HBM) bound: each wave streams 16 rows of W
// straight into registers (never reused), X (16 rows) sits in LDS.
#include "ops.h"

namespace gtav {

namespace {

// One block = 4 waves x FT feature tiles of 16 = 64 FT output features x one slab of 16 RT rows of X.  A wave reads each X fragment from LDS once per 32 k and uses it
// for its FT feature tiles; each W fragment comes straight from memory once and is used for the RT row tiles.  At tens of rows (FT = RT = 1) the launch is a weight
// stream; at hundreds — the adaLN table of a batch-8 frame, 808 rows x 198 656 features — FT = RT = 1 was bound by the LDS read per four MFMAs (11.3 ms = 30 TFLOP/s),
// FT = 4 by re-streaming W_ada once per 16 rows (7.2 ms = 5.9 TB/s out of L2 / Infinity Cache), hence 32-row slabs on top (tools/skinny_bench.py).  Per output the
// same fp32 fma chain in the same order for every FT / RT.
// KC = 0: the whole K of the slab is staged once (RT <= 2 at K = 1024).  KC > 0 (round 4): the slab is staged KC columns at a time, which lets a block keep 80-112 rows
// (RT = 5 .. 7): W_ada is then streamed ONCE for the 80 rows of a training batch / the 101 rows of a batch-1 frame instead of three or four times, and 9 instead of 26
// times for the 808 rows of a batch-8 frame.
template <int ACT, int FT, int RT, int KC = 0>
__global__ __launch_bounds__(256) void skinny_f32_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                        int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [16 RT][(KC ? KC : K) + 8]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.y * 16 * RT;
    const int n0 = (blockIdx.x * 4 + w) * 16 * FT;
    const int kspan = KC ? KC : K;          // columns staged at a time (K % KC == 0, host-checked)
    const int ldk = kspan + 8;
    const bool live = n0 < N;               // (with KC every wave takes part in the staging barriers)

    auto stage = [&](int k0) {              // rows beyond M are zero
        const int kq = kspan >> 2;
        for (int idx = tid; idx < 16 * RT * kq; idx += 256) {
            const int r = idx / kq, c = idx - r * kq;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (m0 + r < M) v = *(const f32x4*)(X + (size_t)(m0 + r) * ldx + k0 + 4 * c);
            *(f32x4*)(xs + r * ldk + 4 * c) = v;
        }
    };
    if constexpr (KC == 0) {
        stage(0);
        __syncthreads();
        if (!live) return;
    }

    const float* wp[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) {
        int nrow = n0 + 16 * f + li;
        nrow = nrow < N ? nrow : N - 1;
        wp[f] = W + (size_t)nrow * K + 4 * g;
    }
    const float* xp = xs + li * ldk + 4 * g;
    f32x4 acc[FT][RT];
#pragma unroll
    for (int f = 0; f < FT; ++f)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // K % 32 == 0; each step covers 32 k: lane group g holds k = kb + 16 i + 4 g + e
    for (int k0 = 0; k0 < K; k0 += kspan) {
    if constexpr (KC != 0) {
        if (k0) __syncthreads();            // every wave has read the chunk before
        stage(k0);
        __syncthreads();
    }
    if (live)
#pragma unroll 2
    for (int kb = 0; kb < kspan; kb += 32) {
        f32x4 x0[RT], x1[RT], w0[FT], w1[FT];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            x0[t] = *(const f32x4*)(xp + (size_t)t * 16 * ldk + kb);
            x1[t] = *(const f32x4*)(xp + (size_t)t * 16 * ldk + kb + 16);
        }
#pragma unroll
        for (int f = 0; f < FT; ++f) {
            w0[f] = *(const f32x4*)(wp[f] + k0 + kb);
            w1[f] = *(const f32x4*)(wp[f] + k0 + kb + 16);
        }
#pragma unroll
        for (int f = 0; f < FT; ++f)
#pragma unroll
            for (int t = 0; t < RT; ++t) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[f][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[f][e], x0[t][e], acc[f][t], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[f][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[f][e], x1[t][e], acc[f][t], 0, 0, 0);
            }
    }
    }
    if (!live) return;
    // D[row = feature 4g + r][col = X row li]
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int m = m0 + 16 * t + li;
#pragma unroll
        for (int f = 0; f < FT; ++f) {
            const int n = n0 + 16 * f + 4 * g;
            if (m < M && n < N) {
                f32x4 v = acc[f][t];
                if (bias) v = v + *(const f32x4*)(bias + n);
                if (ACT == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.0f + expf(-v[e]));
                }
                *(f32x4*)(Y + (size_t)m * ldy + n) = v;
            }
        }
    }
}

}  // namespace

// (FT, RT) variants: (1, 1) small projections; (4, 1) many features, tens of rows (the adaLN table at batch 1); (4, 2) / (8, 2) hundreds of rows
#define GTAV_SKINNY_VARIANTS(X) X(0, 1, 1) X(1, 1, 1) X(0, 4, 1) X(1, 4, 1) X(0, 4, 2) X(1, 4, 2) X(0, 8, 2)
#define GTAV_SKINNY_CHUNKED(X) X(5) X(6) X(7)   // (ACT 0, FT 4, RT, KC 128)
int skinny_init() {
    static unsigned long long done_devs = 0;   // the attribute is per device (see attention.hip)
    int devid = 0;
    GTAV_CHECK_HIP(hipGetDevice(&devid));
    if (done_devs >> (devid & 63) & 1) return 0;
#define GTAV_SKINNY_ATTR(A, F, R) GTAV_CHECK_HIP(hipFuncSetAttribute((const void*)skinny_f32_kernel<A, F, R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    GTAV_SKINNY_VARIANTS(GTAV_SKINNY_ATTR)
#undef GTAV_SKINNY_ATTR
#define GTAV_SKINNY_ATTR_C(R) GTAV_CHECK_HIP(hipFuncSetAttribute((const void*)skinny_f32_kernel<0, 4, R, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    GTAV_SKINNY_CHUNKED(GTAV_SKINNY_ATTR_C)
#undef GTAV_SKINNY_ATTR_C
    done_devs |= 1ull << (devid & 63);
    return 0;
}

int launch_skinny_f32(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int M, int N,
                      int K, int act_silu, hipStream_t stream) {
    GTAV_REQUIRE(M > 0 && N > 0 && N % 4 == 0, "skinny: bad M=%d N=%d", M, N);
    GTAV_REQUIRE(K > 0 && K % 32 == 0 && K <= 2048, "skinny: K=%d must be a multiple of 32 and <= 2048", K);
    GTAV_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ldx >= K && ldy >= N, "skinny: bad leading dims");
    // four (eight) feature tiles per wave where that still leaves the chip full, 32-row slabs as soon as there are two of them: the adaLN table of a frame
    // 1.52 -> 0.92 ms at 101 rows (batch 1), 11.3 -> 5.0 ms at 808 rows (batch 8)
    static const int ft_min_blocks = GTAV_ENV_INT("GTAV_SKINNY_FT_MIN_BLOCKS", 1024);   // experiments build: a huge value = one tile per wave everywhere (A/B)
    static const int variant = GTAV_ENV_INT("GTAV_SKINNY_VARIANT", 0);                  // experiments build: 41 / 42 / 82 force (FT, RT)
    int ft = 1, rt = 1;
    if ((long long)cdiv(N, 256) * cdiv(M, 16) >= ft_min_blocks) {
        ft = 4;
        if (M > 32 && (size_t)32 * (K + 8) * sizeof(float) <= 150 * 1024) {
            rt = 2;
            if (!act_silu && (long long)cdiv(N, 512) * cdiv(M, 32) >= ft_min_blocks) ft = 8;
        }
    }
    // many features, more than two row tiles, no activation (the adaLN table): 80-112 rows per block on K chunks of 128 (44-61 KiB of LDS: two or three blocks per CU) — W is streamed once per 5-7 row tiles
    static const int chunked = GTAV_ENV_INT("GTAV_SKINNY_CHUNKED", 1);   // experiments build: 0 = the whole-K slabs above (A/B)
    if (chunked && !act_silu && M > 32 && K % 128 == 0 && (long long)cdiv(N, 256) >= 256 && variant == 0) {
        const int tiles = cdiv(M, 16);
        const int rtc = tiles <= 5 ? 5 : tiles == 6 ? 6 : tiles == 7 ? 7 : 6;   // one block of rows up to 112 rows, 96-row blocks beyond
        dim3 grid(cdiv(N, 256), cdiv(M, 16 * rtc)), block(256);
        const size_t lds = (size_t)16 * rtc * (128 + 8) * sizeof(float);
#define GTAV_SKINNY_LAUNCH_C(R) if (rtc == R) hipLaunchKernelGGL((skinny_f32_kernel<0, 4, R, 128>), grid, block, lds, stream, X, ldx, W, bias, Y, ldy, M, N, K);
        GTAV_SKINNY_CHUNKED(GTAV_SKINNY_LAUNCH_C)
#undef GTAV_SKINNY_LAUNCH_C
        GTAV_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (variant == 41) ft = 4, rt = 1;
    if (variant == 42) ft = 4, rt = 2;
    if (variant == 82 && !act_silu) ft = 8, rt = 2;
    dim3 grid(cdiv(N, 64 * ft), cdiv(M, 16 * rt)), block(256);
    const size_t lds = (size_t)16 * rt * (K + 8) * sizeof(float);
#define GTAV_SKINNY_LAUNCH(A, F, R)                                                                                               \
    if (act_silu == A && ft == F && rt == R) hipLaunchKernelGGL((skinny_f32_kernel<A, F, R>), grid, block, lds, stream, X, ldx, W, bias, Y, ldy, M, N, K);
    act_silu = act_silu ? 1 : 0;
    GTAV_SKINNY_VARIANTS(GTAV_SKINNY_LAUNCH)
#undef GTAV_SKINNY_LAUNCH
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace gtav
