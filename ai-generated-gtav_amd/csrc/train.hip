// Training-side kernels of the DiT (backward of train_dit.py:680 `accelerator.backward`, AdamW of :232-238,
// gradient clipping of :965-967).  The dense contractions of the backward pass run on the forward's fp16 MFMA GEMM (csrc/gemm.hip)
//   dX = dY W        ->  launch_gemm(X = dY [M][N], W = W^T [K][N])                 (contraction over the layer's outputs)
//   dW = dY^T X      ->  launch_gemm(X = dY^T [N][Mp], W = X^T [K][Mp], EPI_RESID)  (contraction over the tokens, accumulating)
// so what lives here is what surrounds them: tile-major transposes, the LayerNorm + modulate / gate / GELU / RoPE / attention
// backward kernels, the per-frame reductions that produce the adaLN gradients, the fp32 conditioning-path backward and the
// optimizer.  Gradients of activations travel in fp16 (tile-major GEMM operands) scaled by the loss scale; everything that is
// accumulated (weight gradients, the residual-stream gradient, LayerNorm statistics) is fp32.  All kernels are HBM-bound or tiny.
#include "ops.h"

namespace gtav {

namespace {

__device__ __forceinline__ float block_sum(float v, float* red /*[16]*/) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum_dpp(v);
    __syncthreads();                       // red[] may still be read from a previous call
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// ------------------------------------------------------------------------------------------------------------------------
// Tile-major transpose: src logical [R][C] (C % 64 == 0, rows padded to 128) -> dst logical [C][Rp], Rp = round_up(R, 64), rows of
// dst padded to 128 by the caller's allocation; dst columns [R, Rp) are written as zeros (they are the K padding of the dW GEMM:
// both operands must be zero there).  One block = one 64 x 64 sub-tile through LDS, 16-byte accesses on both sides.  (Four sub-tiles per block with every load
// before the first store: 17.2 -> 18.7 us per launch; columns by the transposing LDS read ds_read_b64_tr_b16, which leaves a 16-lane group with 16 different
// destination rows = scattered 16-byte stores: 22.6 us — round 3; the eight-lanes-per-128-byte-row store pattern of this form is what matters.)
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_tiled_kernel(const f16* __restrict__ src, int R, int C, f16* __restrict__ dst, int Rp) {
    __shared__ f16 t[64][72];              // [r][c], 144-byte pitch
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int q = threadIdx.x; q < 512; q += 256) {
        const int r = q >> 3, ch = q & 7;              // row of the sub-tile, 8-element chunk along c
        uint4 v = uint4{0, 0, 0, 0};
        if (r0 + r < R) v = *(const uint4*)(src + tiled_off(r0 + r, c0 + 8 * ch, C));
        *(uint4*)&t[r][8 * ch] = v;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 512; q += 256) {
        const int c = q >> 3, rh = q & 7;              // dst row (= src column), 8-element chunk along r
        union { f16 h[8]; uint4 u; } o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o.h[i] = t[8 * rh + i][c];
        if (r0 + 8 * rh < Rp) *(uint4*)(dst + tiled_off(c0 + c, r0 + 8 * rh, Rp)) = o.u;
    }
}

// fp32 row-major [R][C] (ld = lds) -> fp16 tile-major of the TRANSPOSE, logical [C][Rp] (Rp = round_up(R, 64)); zero padding in
// both directions inside [round_up(C, 128)][Rp].  The W^T operands of the dX GEMMs, refreshed after every optimizer step.
__global__ void convert_T_kernel(const float* __restrict__ src, int lds, int R, int C, f16* __restrict__ dst, int Cp, int Rp) {
    const size_t total = (size_t)Cp * Rp;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx % Rp);     // source row = destination column
        const int c = (int)(idx / Rp);
        float v = 0.f;
        if (r < R && c < C) v = src[(size_t)r * lds + c];
        if (v == v) v = __builtin_amdgcn_fmed3f(v, -F16_MAX, F16_MAX);
        dst[tiled_off(c, r, Rp)] = (f16)v;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// GELU(tanh) forward / backward on flat fp16 buffers (u and h / dh and du share one layout, so tile-major needs no index math).
//   gelu'(u) = s + u s (1 - s) 2 k (1 + 3 c u^2),  s = sigmoid(2 k (u + c u^3)), k = sqrt(2/pi), c = 0.044715
// ------------------------------------------------------------------------------------------------------------------------
__global__ void gelu_tiled_kernel(const f16* __restrict__ u, f16* __restrict__ h, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        union { uint4 q; f16 e[8]; } a, o;
        a.q = ((const uint4*)u)[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) o.e[j] = (f16)__builtin_amdgcn_fmed3f(gelu_tanh_f((float)a.e[j]), -F16_MAX, F16_MAX);
        ((uint4*)h)[i] = o.q;
    }
}
__global__ void gelu_bwd_tiled_kernel(const f16* __restrict__ dh, const f16* __restrict__ u, f16* __restrict__ du, size_t n8, int* err_flag) {
    float amax = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        union { uint4 q; f16 e[8]; } a, b, o;
        a.q = ((const uint4*)dh)[i];
        b.q = ((const uint4*)u)[i];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f32x2_ v = f32x2_{(float)a.e[j], (float)a.e[j + 1]} * gelu_tanh_grad_f2(f32x2_{(float)b.e[j], (float)b.e[j + 1]});
            amax = fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1])));
            o.e[j] = (f16)__builtin_amdgcn_fmed3f(v[0], -F16_MAX, F16_MAX);
            o.e[j + 1] = (f16)__builtin_amdgcn_fmed3f(v[1], -F16_MAX, F16_MAX);
        }
        ((uint4*)du)[i] = o.q;
    }
    sat_report(amax, err_flag);
}

// ------------------------------------------------------------------------------------------------------------------------
// LayerNorm (eps 1e-6, no affine) + modulate backward, one block per token row (D / 4 threads):
//   xhat = (x - mean) rstd,  g = dxn (1 + scale + 1e-6),  dx = rstd (g - mean(g) - xhat mean(g xhat))
//   dres[m] = (accumulate ? dres[m] : 0) + dx;   stats[m] = (mean, rstd) for the per-frame reductions
// The adaLN gradients are sums over the P tokens of a frame: dshift = sum dxn, dscale = sum dxn xhat (frame_reduce_ln_kernel).
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void ln_mod_bwd_kernel(const float* __restrict__ dxn, const float* __restrict__ x, const float* __restrict__ scale,
                                                         int mod_stride, int rows_per_mod, int M, int D, float* __restrict__ dres,
                                                         int accumulate, float* __restrict__ stats) {
    __shared__ float red[16];
    const int m = blockIdx.x, c = threadIdx.x * 4;
    const bool act = c < D;
    const float* sc = scale + (size_t)(m / rows_per_mod) * mod_stride;
    f32x4 xv = f32x4{0.f, 0.f, 0.f, 0.f}, gv = xv, sv = xv;
    if (act) {
        xv = *(const f32x4*)(x + (size_t)m * D + c);
        gv = *(const f32x4*)(dxn + (size_t)m * D + c);
        sv = *(const f32x4*)(sc + c);
    }
    const float mean = block_sum((xv[0] + xv[1]) + (xv[2] + xv[3]), red) / (float)D;
    f32x4 d = xv - mean;
    const float var = block_sum(act ? (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]) : 0.f, red) / (float)D;
    const float rstd = 1.0f / sqrtf(var + 1e-6f);
    f32x4 xh = d * rstd, g;
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = act ? gv[e] * (1.0f + (sv[e] + 1e-6f)) : 0.f;
    const float s1 = block_sum((g[0] + g[1]) + (g[2] + g[3]), red) / (float)D;
    const float s2 = block_sum(act ? (g[0] * xh[0] + g[1] * xh[1]) + (g[2] * xh[2] + g[3] * xh[3]) : 0.f, red) / (float)D;
    if (threadIdx.x == 0) { stats[2 * (size_t)m] = mean; stats[2 * (size_t)m + 1] = rstd; }
    if (!act) return;
    f32x4 dx;
#pragma unroll
    for (int e = 0; e < 4; ++e) dx[e] = rstd * (g[e] - s1 - xh[e] * s2);
    float* o = dres + (size_t)m * D + c;
    if (accumulate) dx = dx + *(const f32x4*)o;
    *(f32x4*)o = dx;
}

// dshift[f][d] = sum over the P tokens of frame f of dxn;  dscale[f][d] = sum of dxn xhat.  One thread per (frame, feature).
__global__ __launch_bounds__(256) void frame_reduce_ln_kernel(const float* __restrict__ dxn, const float* __restrict__ x, const float* __restrict__ stats, int frames,
                                                              int P, int D, float* __restrict__ dshift, float* __restrict__ dscale, int mod_stride) {
    // block = 64 features x 4 token groups (tokens tg, tg + 4, ...): four times the loads in flight of one thread per feature
    __shared__ float ra[4][64], rb[4][64];
    const int dl = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + dl, f = blockIdx.y;
    float a = 0.f, b = 0.f;
    if (d < D)
        for (int t = tg; t < P; t += 4) {
            const size_t m = (size_t)f * P + t;
            const float g = dxn[m * D + d];
            a += g;
            b += g * (x[m * D + d] - stats[2 * m]) * stats[2 * m + 1];
        }
    ra[tg][dl] = a;
    rb[tg][dl] = b;
    __syncthreads();
    if (tg == 0 && d < D) {
        dshift[(size_t)f * mod_stride + d] = (ra[0][dl] + ra[1][dl]) + (ra[2][dl] + ra[3][dl]);
        dscale[(size_t)f * mod_stride + d] = (rb[0][dl] + rb[1][dl]) + (rb[2][dl] + rb[3][dl]);
    }
}

// ln_mod_bwd + frame_reduce_ln in ONE pass over dxn and x (round 4).  Block = 16 token rows of one frame, four waves; a WAVE owns a row (D / 256 float4 per lane,
// DPP reductions: no LDS, no barrier per row) and walks rows w, w + 4, ...; the per-feature sums of the frame's adaLN gradients (dshift = sum dxn, dscale = sum dxn
// xhat) stay in registers over the wave's rows, the four waves are combined through LDS in wave order and the block writes ONE partial row pair; the partial rows of
// a frame are added in chunk order by frame_partials_reduce_kernel.  The pair of kernels it replaces read dxn and x twice.
template <int NV>
__global__ __launch_bounds__(256) void ln_mod_bwd_fused_kernel(const float* __restrict__ dxn, const float* __restrict__ x, const float* __restrict__ scale, int mod_stride,
                                                               int P, float* __restrict__ dres, int accumulate, float* __restrict__ part /*[frames][chunks][2][D]*/) {
    constexpr int D = 256 * NV, RB = 16;
    __shared__ float comb[4][2][D];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int f = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
    f32x4 sc[NV], ash[NV], asc[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        sc[q] = *(const f32x4*)(scale + (size_t)f * mod_stride + 4 * (lane + 64 * q));
        ash[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        asc[q] = ash[q];
    }
    const int t_end = min(P, (chunk + 1) * RB);
    for (int t = chunk * RB + w; t < t_end; t += 4) {
        const size_t m = (size_t)f * P + t;
        f32x4 xv[NV], gv[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            xv[q] = *(const f32x4*)(x + m * D + 4 * (lane + 64 * q));
            gv[q] = *(const f32x4*)(dxn + m * D + 4 * (lane + 64 * q));
        }
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < NV; ++q) s += (xv[q][0] + xv[q][1]) + (xv[q][2] + xv[q][3]);
        const float mean = wave_sum_dpp(s) / (float)D;
        s = 0.f;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            xv[q] = xv[q] - mean;
            s += (xv[q][0] * xv[q][0] + xv[q][1] * xv[q][1]) + (xv[q][2] * xv[q][2] + xv[q][3] * xv[q][3]);
        }
        const float rstd = 1.0f / sqrtf(wave_sum_dpp(s) / (float)D + 1e-6f);
        float s1 = 0.f, s2 = 0.f;
        f32x4 g[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            xv[q] = xv[q] * rstd;                                   // xhat
            ash[q] = ash[q] + gv[q];
            asc[q] = asc[q] + gv[q] * xv[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) g[q][e] = gv[q][e] * (1.0f + (sc[q][e] + 1e-6f));
            s1 += (g[q][0] + g[q][1]) + (g[q][2] + g[q][3]);
            s2 += (g[q][0] * xv[q][0] + g[q][1] * xv[q][1]) + (g[q][2] * xv[q][2] + g[q][3] * xv[q][3]);
        }
        s1 = wave_sum_dpp(s1) / (float)D;
        s2 = wave_sum_dpp(s2) / (float)D;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            float* o = dres + m * D + 4 * (lane + 64 * q);
            f32x4 dx;
#pragma unroll
            for (int e = 0; e < 4; ++e) dx[e] = rstd * (g[q][e] - s1 - xv[q][e] * s2);
            if (accumulate) dx = dx + *(const f32x4*)o;
            *(f32x4*)o = dx;
        }
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        *(f32x4*)(&comb[w][0][4 * (lane + 64 * q)]) = ash[q];
        *(f32x4*)(&comb[w][1][4 * (lane + 64 * q)]) = asc[q];
    }
    __syncthreads();
    float* out = part + ((size_t)f * chunks + chunk) * 2 * D;
    for (int i = threadIdx.x; i < 2 * D; i += 256) {
        const int k = i / D, n = i - k * D;
        out[i] = (comb[0][k][n] + comb[1][k][n]) + (comb[2][k][n] + comb[3][k][n]);
    }
}
// dshift[f][n] / dscale[f][n] = the frame's partial rows added in chunk order
__global__ __launch_bounds__(256) void frame_partials_reduce_kernel(const float* __restrict__ part, int chunks, int D, float* __restrict__ dshift, float* __restrict__ dscale, int mod_stride) {
    const int n = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (n >= D) return;
    float a = 0.f, b = 0.f;
    for (int c = 0; c < chunks; ++c) {
        a += part[(((size_t)f * chunks + c) * 2 + 0) * D + n];
        b += part[(((size_t)f * chunks + c) * 2 + 1) * D + n];
    }
    dshift[(size_t)f * mod_stride + n] = a;
    dscale[(size_t)f * mod_stride + n] = b;
}

// Gated residual branch x += gate y (model/dit.py:207-223) backward: dy = gate dres -> fp16 TILE-MAJOR (the dX / dW GEMM operand);
// dgate[f][d] = sum over the frame's tokens of dres y (frame_reduce_gate_kernel; y = the branch output saved by the forward).
__global__ void gate_bwd_kernel(const float* __restrict__ dres, const float* __restrict__ gate, int mod_stride, int rows_per_mod, int M, int D,
                                f16* __restrict__ dy, int* err_flag) {
    const size_t total = (size_t)M * (D / 4);
    float amax = 0.f;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % (D / 4)) * 4;
        const int m = (int)(idx / (D / 4));
        f32x4 v = *(const f32x4*)(dres + (size_t)m * D + c);
        if (gate) v = v * *(const f32x4*)(gate + (size_t)(m / rows_per_mod) * mod_stride + c);
        *(f16x4*)(dy + tiled_off(m, c, D)) = sat4(v[0], v[1], v[2], v[3], amax);
    }
    sat_report(amax, err_flag);
}
__global__ __launch_bounds__(256) void frame_reduce_gate_kernel(const float* __restrict__ dres, const f16* __restrict__ y, int frames, int P, int D,
                                                                float* __restrict__ dgate, int mod_stride) {
    __shared__ float ra[4][64];
    const int dl = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + dl, f = blockIdx.y;
    float a = 0.f;
    if (d < D)
        for (int t = tg; t < P; t += 4) {
            const size_t m = (size_t)f * P + t;
            a += dres[m * D + d] * (float)y[m * D + d];
        }
    ra[tg][dl] = a;
    __syncthreads();
    if (tg == 0 && d < D) dgate[(size_t)f * mod_stride + d] = (ra[0][dl] + ra[1][dl]) + (ra[2][dl] + ra[3][dl]);
}

// gelu_bwd_tiled + the column sums of its output (the fc1 bias gradient) in one pass (round 4): block = one 64-column tile column x a slice of rows (the layout of
// colsum_tiled_kernel), thread = (8-column chunk, row lane).  du = sat16(dh gelu'(u)) for every row of the padded image; the sums take the unrounded products of the
// M real rows: ws[split][N], added in split order by colsum_reduce_kernel.
__global__ __launch_bounds__(256) void gelu_bwd_colsum_kernel(const f16* __restrict__ dh, const f16* __restrict__ u, f16* __restrict__ du, int M, int Mp, int N,
                                                              int rows_per_block, float* __restrict__ ws, int* err_flag) {
    __shared__ float part[32][65];
    const int ch = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int n0 = blockIdx.x * 64;
    const int r_begin = blockIdx.y * rows_per_block, r_end = min(Mp, r_begin + rows_per_block);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float amax = 0.f;
    // two rows per trip, all four loads first: the loop is a dependent load -> 40 issue slots -> store chain per row otherwise (5.0 -> 5.6 TB/s, round 5)
    for (int r = r_begin + rl; r < r_end; r += 64) {
        const int r2 = r + 32;
        const bool has2 = r2 < r_end;
        const size_t off = tiled_off(r, n0 + 8 * ch, N), off2 = tiled_off(has2 ? r2 : r, n0 + 8 * ch, N);
        union U8 { uint4 q; f16 e[8]; };
        U8 x[2], b[2], o[2];
        x[0].q = *(const uint4*)(dh + off);
        b[0].q = *(const uint4*)(u + off);
        x[1].q = *(const uint4*)(dh + off2);
        b[1].q = *(const uint4*)(u + off2);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bool real = (t == 0 ? r : r2) < M && (t == 0 || has2);   // (pad rows of the 128-row image are transformed like the others but must not reach the sums — they may hold anything, NaN included)
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const f32x2_ v = f32x2_{(float)x[t].e[j], (float)x[t].e[j + 1]} * gelu_tanh_grad_f2(f32x2_{(float)b[t].e[j], (float)b[t].e[j + 1]});
                amax = fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1])));
                o[t].e[j] = (f16)__builtin_amdgcn_fmed3f(v[0], -F16_MAX, F16_MAX);
                o[t].e[j + 1] = (f16)__builtin_amdgcn_fmed3f(v[1], -F16_MAX, F16_MAX);
                a[j] += real ? v[0] : 0.f;
                a[j + 1] += real ? v[1] : 0.f;
            }
        }
        *(uint4*)(du + off) = o[0].q;
        if (has2) *(uint4*)(du + off2) = o[1].q;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) part[rl][8 * ch + i] = a[i];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
        for (int i = 0; i < 32; ++i) t += part[i][threadIdx.x];
        ws[(size_t)blockIdx.y * N + n0 + threadIdx.x] = t;
    }
    sat_report(amax, err_flag);
}

// gate_bwd + frame_reduce_gate + the bias gradient of the Linear in front of the gate in ONE pass over dres (round 4): block = 64 features of one frame,
// 16 feature quads x 16 token lanes.  dy = sat16(gate dres) tile-major; dgate[f][n] = sum_t dres y; bias_part[f][n] = gate[f][n] sum_t dres (= the frame's share of
// sum_m dy[m][n], from the unrounded products; colsum_reduce_kernel adds the frames in order).  The three kernels it replaces read dres twice and dy once more.
__global__ __launch_bounds__(256) void gate_bwd_fused_kernel(const float* __restrict__ dres, const f16* __restrict__ y, const float* __restrict__ gate, int mod_stride,
                                                             int P, int D, f16* __restrict__ dy, float* __restrict__ dgate, float* __restrict__ bias_part, int* err_flag) {
    __shared__ float ra[16][65], rs[16][65];
    const int qd = threadIdx.x & 15, tl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + 4 * qd, f = blockIdx.y;
    f32x4 gt = f32x4{1.f, 1.f, 1.f, 1.f};
    if (gate) gt = *(const f32x4*)(gate + (size_t)f * mod_stride + c);
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, sm = a;
    float amax = 0.f;
    for (int t = tl; t < P; t += 16) {
        const size_t m = (size_t)f * P + t;
        const f32x4 v = *(const f32x4*)(dres + m * D + c);
        const f16x4 yy = *(const f16x4*)(y + m * D + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += v[e] * (float)yy[e];
        sm = sm + v;
        const f32x4 o = v * gt;
        *(f16x4*)(dy + tiled_off((int)m, c, D)) = sat4(o[0], o[1], o[2], o[3], amax);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { ra[tl][4 * qd + e] = a[e]; rs[tl][4 * qd + e] = sm[e]; }
    __syncthreads();
    if (threadIdx.x < 64) {
        float ta = 0.f, ts = 0.f;
        for (int i = 0; i < 16; ++i) { ta += ra[i][threadIdx.x]; ts += rs[i][threadIdx.x]; }
        const int n = blockIdx.x * 64 + threadIdx.x;
        dgate[(size_t)f * mod_stride + n] = ta;
        if (bias_part) bias_part[(size_t)f * D + n] = (gate ? gate[(size_t)f * mod_stride + n] : 1.0f) * ts;
    }
    sat_report(amax, err_flag);
}

// db[n] += sum_m dY[m][n] for a tile-major fp16 dY (logical [M][N], N % 64 == 0).  Block = one 64-column tile column x a slice of rows;
// thread = (8-column chunk, row lane): one 16-byte load per row.
// Row splits: with `ws` the per-split column sums go to ws[split][N] and colsum_reduce_kernel adds them to db in split order — a fixed
// summation order (bit-reproducible gradients: the resume test compares weights bit for bit); without it (one split) the block adds directly.
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ ws, int splits, int N, float* __restrict__ db) {
    // block = 32 columns x 8 split lanes: lane j of a column adds splits j, j + 8, ... (independent loads), the eight lane sums are added in lane order through LDS.
    // (one thread per column made this launch `splits` dependent memory round trips on 4-16 blocks: 7 us for a few hundred KB)
    __shared__ float part[8][33];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + cl;
    float t = 0.f;
    if (n < N)
        for (int sp = sl; sp < splits; sp += 8) t += ws[(size_t)sp * N + n];
    part[sl][cl] = t;
    __syncthreads();
    if (sl == 0 && n < N) db[n] += ((part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl])) + ((part[4][cl] + part[5][cl]) + (part[6][cl] + part[7][cl]));
}
// up to four column-sum reductions in ONE launch (blockIdx.y = job): the three bias gradients of a half-block (fc2, fc1, out-projection) each paid a launch of their own
struct ColsumJobs { const float* ws[4]; float* db[4]; int splits[4]; int N[4]; };
__global__ __launch_bounds__(256) void colsum_reduce_multi_kernel(ColsumJobs jobs) {
    __shared__ float part[8][33];
    const int j = blockIdx.y;
    const float* ws = jobs.ws[j];
    const int N = jobs.N[j], splits = jobs.splits[j];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + cl;
    if (blockIdx.x * 32 >= N) return;      // (block-uniform)
    float t = 0.f;
    if (n < N)
        for (int sp = sl; sp < splits; sp += 8) t += ws[(size_t)sp * N + n];
    part[sl][cl] = t;
    __syncthreads();
    if (sl == 0 && n < N) jobs.db[j][n] += ((part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl])) + ((part[4][cl] + part[5][cl]) + (part[6][cl] + part[7][cl]));
}
__global__ __launch_bounds__(256) void colsum_tiled_kernel(const f16* __restrict__ dy, int M, int N, float* __restrict__ db, int rows_per_block, float* __restrict__ ws) {
    __shared__ float part[32][65];
    const int ch = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int n0 = blockIdx.x * 64;
    const int r_begin = blockIdx.y * rows_per_block, r_end = min(M, r_begin + rows_per_block);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = r_begin + rl; r < r_end; r += 32) {
        union { uint4 q; f16 e[8]; } v;
        v.q = *(const uint4*)(dy + tiled_off(r, n0 + 8 * ch, N));
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] += (float)v.e[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) part[rl][8 * ch + i] = a[i];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
        for (int i = 0; i < 32; ++i) t += part[i][threadIdx.x];
        if (ws) ws[(size_t)blockIdx.y * N + n0 + threadIdx.x] = t;
        else db[n0 + threadIdx.x] += t;      // single split: the only block of this column tile
    }
}
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ a, int lda, int M, int N, float* __restrict__ db, int rows_per_block, float* __restrict__ ws) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int r_begin = blockIdx.y * rows_per_block, r_end = min(M, r_begin + rows_per_block);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // (four independent chains, as colsum_reduce_kernel)
    int r = r_begin;
    for (; r + 4 <= r_end; r += 4) {
        s0 += a[(size_t)r * lda + n];
        s1 += a[(size_t)(r + 1) * lda + n];
        s2 += a[(size_t)(r + 2) * lda + n];
        s3 += a[(size_t)(r + 3) * lda + n];
    }
    for (; r < r_end; ++r) s0 += a[(size_t)r * lda + n];
    const float s = (s0 + s1) + (s2 + s3);
    if (ws) ws[(size_t)blockIdx.y * N + n] = s;
    else db[n] += s;
}

// fp32 row-major [M][D] -> fp16 tile-major (the patch-embedding gradient as a dW operand)
__global__ void to_tiled_f16_kernel(const float* __restrict__ a, int M, int D, f16* __restrict__ out, int* err_flag) {
    const size_t total = (size_t)M * (D / 4);
    float amax = 0.f;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % (D / 4)) * 4;
        const int m = (int)(idx / (D / 4));
        const f32x4 v = *(const f32x4*)(a + (size_t)m * D + c);
        *(f16x4*)(out + tiled_off(m, c, D)) = sat4(v[0], v[1], v[2], v[3], amax);
    }
    sat_report(amax, err_flag);
}

// Loss gradient: loss = mean((v_pred[:, -1] - v_target)^2) over B n elements (train_dit.py:650).  d v_pred is non-zero for the last
// frame only; it goes straight to the final projection's output gradient dfo[m][f] (m = token, f = (ph, pw, c): the inverse of
// DiT.unpatchify, model/dit.py:328-341) as an fp16 tile-major GEMM operand, times `scale` = 2 loss_scale / (B n).
__global__ void mse_bwd_patch_kernel(const float* __restrict__ vpred, const float* __restrict__ vtarget, int B, int T, int C, int H, int W, int p,
                                     float scale, f16* __restrict__ dfo, int ldf, int* err_flag) {
    const int gh = H / p, gw = W / p, F = C * p * p;
    const size_t total = (size_t)B * T * gh * gw * ldf;
    float amax = 0.f;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(idx % ldf);
        const size_t m = idx / ldf;
        const int gx = (int)(m % gw), gy = (int)((m / gw) % gh), t = (int)((m / ((size_t)gw * gh)) % T), b = (int)(m / ((size_t)gw * gh * T));
        float v = 0.f;
        if (t == T - 1 && f < F) {
            const int c = f % C, pw = (f / C) % p, ph = f / (C * p);
            const size_t pix = ((size_t)c * H + (gy * p + ph)) * W + (gx * p + pw);
            v = scale * (vpred[((size_t)b * T + t) * C * H * W + pix] - vtarget[(size_t)b * C * H * W + pix]);
        }
        amax = fmaxf(amax, fabsf(v));
        dfo[tiled_off((int)m, f, ldf)] = (f16)__builtin_amdgcn_fmed3f(v, -F16_MAX, F16_MAX);
    }
    sat_report(amax, err_flag);
}

// ------------------------------------------------------------------------------------------------------------------------
// Spatial attention backward (model/attention.py:99-136), one block per (frame, head), S tokens, head_dim 64, fp32 math.
//   P = softmax(Q K^T / 8);  dV = P^T dO;  dP = dO V^T;  dS = P (dP - rowsum(P dP)) / 8;  dQ = dS K;  dK = dS^T Q
// Inputs in the forward's layouts (Q, K [nb][head][S][64] with RoPE applied, Vt [nb][head][64][S]); dO fp16 row-major [M][D].
// Output: dqkv fp16 TILE-MAJOR logical [M][3 D] (q | k | v column blocks of the to_qkv Linear), dq / dk rotated back through the
// interleaved-pair RoPE (the transpose of a rotation is the rotation by the negative angle).
// Query rows are processed 16 at a time: their P and dS rows live in LDS, dK / dV accumulate in registers (thread = (d, key group)).
// ------------------------------------------------------------------------------------------------------------------------
constexpr int AB_MAXS = 160;   // S <= 160 (DiT: 144)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot8(const f16x8& a, const f16x8& b, float acc) {
    // v_dot2_f32_f16: two fp16 products accumulated in fp32 per instruction
    acc = dot2acc(f16x2{a[0], a[1]}, f16x2{b[0], b[1]}, acc, false);
    acc = dot2acc(f16x2{a[2], a[3]}, f16x2{b[2], b[3]}, acc, false);
    acc = dot2acc(f16x2{a[4], a[5]}, f16x2{b[4], b[5]}, acc, false);
    acc = dot2acc(f16x2{a[6], a[7]}, f16x2{b[6], b[7]}, acc, false);
    return acc;
}
template <int NT>   // threads per block: NT / 16 query rows per row block, NT / 64 key groups
__global__ __launch_bounds__(NT) void attn_spatial_bwd_kernel(const f16* __restrict__ Q, const f16* __restrict__ K, const f16* __restrict__ Vt,
                                                              const f16* __restrict__ dO, int heads, int S, int D,
                                                              const float* __restrict__ rope_cs, f16* __restrict__ dqkv, int* err_flag) {
    constexpr int RB = NT / 16, JG = NT / 64;
    extern __shared__ __attribute__((aligned(16))) char smraw[];
    // rows of 64 halves padded to 72 (144 bytes): a 16-byte read of consecutive rows by consecutive lanes is bank-conflict free
    constexpr int LP = 72;
    f16* sQ = (f16*)smraw;                 // [S][LP]
    f16* sK = sQ + S * LP;
    f16* sV = sK + S * LP;                 // [S][LP] (transposed back from Vt)
    f16* sdO = sV + S * LP;
    float* sP = (float*)(sdO + S * LP);    // [RB][SP]
    const int SP = S + 4;                  // fp32 row pitch of the P / dS row blocks
    float* sdS = sP + RB * SP;
    const int item = blockIdx.x, nb = item / heads, head = item % heads;
    const int tid = threadIdx.x;
    const f16* q = Q + (size_t)item * S * 64;
    const f16* k = K + (size_t)item * S * 64;
    const f16* vt = Vt + (size_t)item * 64 * S;
    for (int i = tid; i < S * 8; i += NT) {        // 16-byte chunks
        const int s = i >> 3, ch = i & 7;
        *(uint4*)(sQ + s * LP + 8 * ch) = ((const uint4*)q)[i];
        *(uint4*)(sK + s * LP + 8 * ch) = ((const uint4*)k)[i];
        *(uint4*)(sdO + s * LP + 8 * ch) = *(const uint4*)(dO + ((size_t)nb * S + s) * D + head * 64 + 8 * ch);
    }
    for (int i = tid; i < S * 64; i += NT) {
        const int d = i / S, s = i % S;
        sV[s * LP + d] = vt[i];
    }
    __syncthreads();
    // dK / dV ownership: feature d_own, keys j = 4 JG c + 4 jg + e (e < 4): the P / dS rows are read as float4
    const int d_own = tid & 63, jg = tid >> 6;
    constexpr int CMAX = (AB_MAXS + 4 * JG - 1) / (4 * JG);
    f32x4 accK[CMAX], accV[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) accK[c] = accV[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ri = tid >> 4, cj = tid & 15;        // score ownership: query row ri of the RB-row block, keys cj, cj + 16, ...
    float amax = 0.f;
    for (int r0 = 0; r0 < S; r0 += RB) {
        const int i = r0 + ri;
        const bool iv = i < S;
        constexpr int NJ = AB_MAXS / 16;
        float sc[NJ], dp[NJ];
        float mx = -INFINITY;
        {
            // this thread's query row and its output-gradient row stay in registers for the whole block of keys
            f16x8 qr[8], gr[8];
            const int ic = iv ? i : S - 1;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                qr[ch] = *(const f16x8*)(sQ + ic * LP + 8 * ch);
                gr[ch] = *(const f16x8*)(sdO + ic * LP + 8 * ch);
            }
#pragma unroll
            for (int c = 0; c < NJ; ++c) {
                const int j = cj + 16 * c;
                sc[c] = -INFINITY;
                dp[c] = 0.f;
                if (j < S) {
                    float a = 0.f, b = 0.f;
#pragma unroll
                    for (int ch = 0; ch < 8; ++ch) {
                        a = dot8(qr[ch], *(const f16x8*)(sK + j * LP + 8 * ch), a);
                        b = dot8(gr[ch], *(const f16x8*)(sV + j * LP + 8 * ch), b);
                    }
                    if (iv) {
                        sc[c] = a * 0.125f;
                        dp[c] = b;
                        mx = fmaxf(mx, sc[c]);
                    }
                }
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < NJ; ++c) {
            sc[c] = (cj + 16 * c < S && iv) ? __expf(sc[c] - mx) : 0.f;
            sum += sc[c];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;
        float dsum = 0.f;
#pragma unroll
        for (int c = 0; c < NJ; ++c) {
            sc[c] *= inv;
            dsum += sc[c] * dp[c];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) dsum += __shfl_xor(dsum, o, 64);
#pragma unroll
        for (int c = 0; c < NJ; ++c) {
            const int j = cj + 16 * c;
            if (j < S) {       // rows beyond S hold zeros (sc = 0): the dK / dV loops below may read all 16 rows
                sP[ri * SP + j] = sc[c];
                sdS[ri * SP + j] = sc[c] * (dp[c] - dsum) * 0.125f;
            }
        }
        __syncthreads();
        // dQ rows of this block: 16 x 64 outputs, 4 per thread (row ri, features 4 cj .. 4 cj + 3); RoPE^T; store
        if (iv) {
            float dq[4] = {0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < S; j += 4) {
                const f32x4 w4 = *(const f32x4*)(sdS + ri * SP + j);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const f16x4 kx = *(const f16x4*)(sK + (j + jj) * LP + 4 * cj);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dq[e] += w4[jj] * (float)kx[e];
                }
            }
            const f32x4 cs = *(const f32x4*)(rope_cs + (size_t)i * 64 + 4 * cj);
            const float o0 = dq[0] * cs[0] + dq[1] * cs[1], o1 = dq[1] * cs[0] - dq[0] * cs[1];
            const float o2 = dq[2] * cs[2] + dq[3] * cs[3], o3 = dq[3] * cs[2] - dq[2] * cs[3];
            *(f16x4*)(dqkv + tiled_off(nb * S + i, head * 64 + 4 * cj, 3 * D)) = sat4(o0, o1, o2, o3, amax);
        }
        // dV[j][d] += sum_r P[r][j] dO[r][d];  dK[j][d] += sum_r dS[r][j] Q[r][d]   (rows past S contribute zeros)
        const int rows = min(RB, S - r0);
        for (int r = 0; r < rows; ++r) {
            const float go = (float)sdO[(r0 + r) * LP + d_own], qq = (float)sQ[(r0 + r) * LP + d_own];
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                const int j = 4 * JG * c + 4 * jg;
                if (j < S) {
                    accV[c] += *(const f32x4*)(sP + r * SP + j) * go;
                    accK[c] += *(const f32x4*)(sdS + r * SP + j) * qq;
                }
            }
        }
        __syncthreads();
    }
    // dK (RoPE^T needs the pair partner: lanes d and d ^ 1 are neighbours in the wave) and dV
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * JG * c + 4 * jg + e;
            if (j < S) {      // uniform per wave: j depends on jg (= the wave index), c and e only
                const float mine = accK[c][e], other = __shfl_xor(mine, 1, 64);
                const float co = rope_cs[(size_t)j * 64 + (d_own & ~1)], si = rope_cs[(size_t)j * 64 + (d_own | 1)];
                const float dk = (d_own & 1) ? mine * co - other * si : mine * co + other * si;
                const size_t m = (size_t)nb * S + j;
                amax = fmaxf(amax, fmaxf(fabsf(dk), fabsf(accV[c][e])));
                dqkv[tiled_off((int)m, D + head * 64 + d_own, 3 * D)] = (f16)__builtin_amdgcn_fmed3f(dk, -F16_MAX, F16_MAX);
                dqkv[tiled_off((int)m, 2 * D + head * 64 + d_own, 3 * D)] = (f16)__builtin_amdgcn_fmed3f(accV[c][e], -F16_MAX, F16_MAX);
            }
        }
    }
    sat_report(amax, err_flag);
}

// ------------------------------------------------------------------------------------------------------------------------
// The same on the matrix cores (the default; the VALU kernel above stays for the experiments build's A/B runs).  One block of nine
// waves (one query / key tile each at S = 144) per (frame, head); Q, K, V, dO as fp16 rows in LDS (pitch 72 halves), 16 x 16 tiles (S = 144: 9 tiles).
//   query-tile pass (wave w: query tiles w, w + 9, ...): S_i = Q_i K^T and dP_i = dO_i V^T for the whole key row (v_mfma_f32_16x16x32_f16:
//     lane (g = l / 16, c = l % 16) holds rows 4 g + r, column c), row statistics over the 16 lanes of a group -> lse2[q] = max c + log2 sum
//     (c = log2(e) / 8) and Dq[q] = sum_j P dP to LDS; then, with the row still in registers, dS_ij = P (dP - Dq) / 8 per key tile and
//     dQ_i^T += K_j^T dS_ij^T with the K = 16 MFMA: K_j^T is a transposing LDS read (ds_read_b64_tr_b16) of the row-major K image, dS^T goes
//     through a private [key][query] scratch (8 bytes per lane out, one transposing read back).  One wave owns the whole dQ row: no
//     cross-wave accumulation (a first version added dQ tiles from the key-tile pass into an fp32 LDS image with ds_add_f32: 650 us per
//     launch, 500 of them in the atomics);
//   key-tile pass (wave w: key tiles j = w, w + 9, ...; dK_j / dV_j in registers): for every query tile i recompute S_ij, dP_ij (2 + 2
//     MFMAs), P = exp2(S c - lse2), dS.  The accumulator layout of a 16 x 16 tile IS the A-operand layout of its transpose for the K = 16
//     MFMA (lane: row c, k = 4 g + r), so dV_j += P^T dO_i and dK_j += dS^T Q_i take P / dS straight from registers; their B operands (4
//     query rows x 16 features, a column gather) are transposing LDS reads of the row-major dO / Q images;
//   epilogue: RoPE^T (the transpose of a rotation is the rotation by the negative angle), dK_j / dV_j into the LDS rows of K_j / V_j (no other
//     wave reads them), dQ from the fp32 image, then 16-byte stores into the tile-major [M][3 D] gradient of the to_qkv output.
// fp16 rounding of P and dS (the MFMA operands) is the only arithmetic difference to the VALU kernel.
// ------------------------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x4 lds_read_tr(const f16* p) {   // lane 4 q + p' of a 16-lane group passes row q, columns 4 p' ..; lane i gets column i of the 4 rows
    return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p));
}
constexpr int ABM_LP = 72, ABM_DQP = 68, ABM_NW = 9, ABM_MAXT = AB_MAXS / 16;   // 9 waves: one query / key tile each at S = 144 (3 waves: 170 us per launch of 1280 items, 9 waves: 99 us)
template <int DBG>   // experiments: 1 = no dQ atomics, 2 = no K = 16 MFMAs / transposed reads, 4 = no RoPE / LDS write-back of dK, dV (timing only)
__global__ __launch_bounds__(64 * ABM_NW) void attn_spatial_bwd_mfma_kernel(const f16* __restrict__ Q, const f16* __restrict__ K, const f16* __restrict__ Vt,
                                                                            const f16* __restrict__ dO, int heads, int S, int D,
                                                                            const float* __restrict__ rope_cs, f16* __restrict__ dqkv, int* err_flag) {
    constexpr int LP = ABM_LP, DQP = ABM_DQP, NW = ABM_NW, NT = 64 * ABM_NW;
    extern __shared__ __attribute__((aligned(16))) char smraw[];
    f16* sQ = (f16*)smraw;                 // [S][LP]
    f16* sK = sQ + S * LP;
    f16* sV = sK + S * LP;
    f16* sdO = sV + S * LP;
    float* sdQ = (float*)(sdO + S * LP);   // [S][DQP]
    float* slse = sdQ + S * DQP;           // [S]
    float* sDq = slse + S;                 // [S]
    f16* sscr = (f16*)(sDq + S);           // [NW][16][16]: dS tile of the wave as [key][query]
    const int item = blockIdx.x, nb = item / heads, head = item % heads;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, g = lane >> 4;
    const int nt = S >> 4;
    const f16* q = Q + (size_t)item * S * 64;
    const f16* k = K + (size_t)item * S * 64;
    const f16* vt = Vt + (size_t)item * 64 * S;
    for (int i = tid; i < S * 8; i += NT) {        // 16-byte chunks
        const int srow = i >> 3, ch = i & 7;
        *(uint4*)(sQ + srow * LP + 8 * ch) = ((const uint4*)q)[i];
        *(uint4*)(sK + srow * LP + 8 * ch) = ((const uint4*)k)[i];
        *(uint4*)(sdO + srow * LP + 8 * ch) = *(const uint4*)(dO + ((size_t)nb * S + srow) * D + head * 64 + 8 * ch);
    }
    for (int i = tid; i < S * 8; i += NT) {        // V^T rows (one feature, S tokens) in 8-token chunks -> [token][feature]; lanes along the feature: conflict-free 2-byte LDS stores
        const int d = i & 63, s0 = (i >> 6) << 3;
        const f16x8 v8 = *(const f16x8*)(vt + (size_t)d * S + s0);
#pragma unroll
        for (int e = 0; e < 8; ++e) sV[(s0 + e) * LP + d] = v8[e];
    }
    __syncthreads();
    const float cexp = 0.125f * 1.4426950408889634f;
    // ---- pre-pass: row statistics ----
    for (int i = w; i < nt; i += NW) {
        f16x8 qa[2], ga[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            qa[h] = *(const f16x8*)(sQ + (16 * i + c) * LP + 32 * h + 8 * g);
            ga[h] = *(const f16x8*)(sdO + (16 * i + c) * LP + 32 * h + 8 * g);
        }
        f32x4 sa[ABM_MAXT], da[ABM_MAXT];
#pragma unroll
        for (int j = 0; j < ABM_MAXT; ++j) {
            sa[j] = da[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < nt) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f16x8 kb = *(const f16x8*)(sK + (16 * j + c) * LP + 32 * h + 8 * g);
                    const f16x8 vb = *(const f16x8*)(sV + (16 * j + c) * LP + 32 * h + 8 * g);
                    sa[j] = mfma16(qa[h], kb, sa[j], 0, 0, 0);
                    da[j] = mfma16(ga[h], vb, da[j], 0, 0, 0);
                }
            }
        }
        f32x4 mx = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int j = 0; j < ABM_MAXT; ++j)
            if (j < nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], sa[j][r]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], o, 64));
        f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f}, pd = sum;
#pragma unroll
        for (int j = 0; j < ABM_MAXT; ++j)
            if (j < nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pe = __builtin_amdgcn_exp2f((sa[j][r] - mx[r]) * cexp);
                    sum[r] += pe;
                    pd[r] = __builtin_fmaf(pe, da[j][r], pd[r]);
                }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sum[r] += __shfl_xor(sum[r], o, 64);
                pd[r] += __shfl_xor(pd[r], o, 64);
            }
        f32x4 lse, dqr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            lse[r] = mx[r] * cexp + __builtin_amdgcn_logf(sum[r]);   // v_log_f32 = log2
            dqr[r] = pd[r] / sum[r];
        }
        if (c == 0) {
            *(f32x4*)(slse + 16 * i + 4 * g) = lse;
            *(f32x4*)(sDq + 16 * i + 4 * g) = dqr;
        }
        // dQ_i^T[feature][query] = sum_j K_j^T dS_ij^T while the whole row of S_i / dP_i is still in registers (one wave owns the query tile:
        // no cross-wave accumulation; an fp32 LDS image fed by ds_add_f32 from the key-tile loop cost 500 us of a 650 us launch)
        f32x4 dqa[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) dqa[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
        f16* scr0 = sscr + w * 256;
#pragma unroll
        for (int j = 0; j < ABM_MAXT; ++j) {
            if (j < nt) {
                f16x4 dsh;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pe = __builtin_amdgcn_exp2f(sa[j][r] * cexp - lse[r]);
                    dsh[r] = (f16)__builtin_amdgcn_fmed3f(pe * (da[j][r] - dqr[r]) * 0.125f, -F16_MAX, F16_MAX);
                }
                *(f16x4*)(scr0 + c * 16 + 4 * g) = dsh;                                       // scratch[key c][queries 4 g ..]
                const f16x4 dsT = lds_read_tr(scr0 + (4 * g + (c >> 2)) * 16 + 4 * (c & 3));    // B[k = key 4 g + e][col = query c]
                if (!(DBG & 2)) {
#pragma unroll
                    for (int ft = 0; ft < 4; ++ft) {
                        const f16x4 kT = lds_read_tr(sK + (16 * j + 4 * g + (c >> 2)) * LP + 16 * ft + 4 * (c & 3));   // A[row = feature 16 ft + c][k = key 4 g + e]
                        dqa[ft] = __builtin_amdgcn_mfma_f32_16x16x16f16(kT, dsT, dqa[ft], 0, 0, 0);           // rows: features 16 ft + 4 g + r; column: query c
                    }
                }
            }
        }
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) *(f32x4*)(sdQ + (16 * i + c) * DQP + 16 * ft + 4 * g) = dqa[ft];
    }
    __syncthreads();
    // ---- main: key tiles of this wave ----
    float amax = 0.f;
    for (int j = w; j < nt; j += NW) {
        f16x8 kb[2], vb[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            kb[h] = *(const f16x8*)(sK + (16 * j + c) * LP + 32 * h + 8 * g);
            vb[h] = *(const f16x8*)(sV + (16 * j + c) * LP + 32 * h + 8 * g);
        }
        f32x4 dKa[4], dVa[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) dKa[ft] = dVa[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < nt; ++i) {
            f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, da = sa;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f16x8 qa = *(const f16x8*)(sQ + (16 * i + c) * LP + 32 * h + 8 * g);
                const f16x8 ga = *(const f16x8*)(sdO + (16 * i + c) * LP + 32 * h + 8 * g);
                sa = mfma16(qa, kb[h], sa, 0, 0, 0);     // rows: queries 16 i + 4 g + r; column: key 16 j + c
                da = mfma16(ga, vb[h], da, 0, 0, 0);
            }
            const f32x4 l4 = *(const f32x4*)(slse + 16 * i + 4 * g), d4 = *(const f32x4*)(sDq + 16 * i + 4 * g);
            f16x4 ph, dsh;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pe = __builtin_amdgcn_exp2f(sa[r] * cexp - l4[r]);
                const float dsv = pe * (da[r] - d4[r]) * 0.125f;
                amax = fmaxf(amax, fabsf(dsv));
                ph[r] = (f16)pe;
                dsh[r] = (f16)__builtin_amdgcn_fmed3f(dsv, -F16_MAX, F16_MAX);
            }
            if (DBG & 2) { amax = fmaxf(amax, (float)ph[0] + (float)dsh[0]); continue; }
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                const f16x4 goT = lds_read_tr(sdO + (16 * i + 4 * g + (c >> 2)) * LP + 16 * ft + 4 * (c & 3));   // B[k = query 4 g + e][col = feature 16 ft + c]
                const f16x4 qT = lds_read_tr(sQ + (16 * i + 4 * g + (c >> 2)) * LP + 16 * ft + 4 * (c & 3));
                dVa[ft] = __builtin_amdgcn_mfma_f32_16x16x16f16(ph, goT, dVa[ft], 0, 0, 0);     // rows: keys 16 j + 4 g + r; column: feature 16 ft + c
                dKa[ft] = __builtin_amdgcn_mfma_f32_16x16x16f16(dsh, qT, dKa[ft], 0, 0, 0);
            }
        }
        // dK_j (RoPE^T: the pair partner of feature 16 ft + c is the neighbouring lane c ^ 1) and dV_j -> LDS rows of K_j / V_j
        if (DBG & 4) { amax = fmaxf(amax, dKa[0][0] + dVa[1][1]); continue; }
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            const int d = 16 * ft + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * j + 4 * g + r;
                const float mine = dKa[ft][r], other = __shfl_xor(mine, 1, 64);
                const float co = rope_cs[(size_t)key * 64 + (d & ~1)], si = rope_cs[(size_t)key * 64 + (d | 1)];
                const float dk = (d & 1) ? mine * co - other * si : mine * co + other * si;
                amax = fmaxf(amax, fmaxf(fabsf(dk), fabsf(dVa[ft][r])));
                sK[key * LP + d] = (f16)__builtin_amdgcn_fmed3f(dk, -F16_MAX, F16_MAX);
                sV[key * LP + d] = (f16)__builtin_amdgcn_fmed3f(dVa[ft][r], -F16_MAX, F16_MAX);
            }
        }
    }
    __syncthreads();
    // ---- epilogue: 16-byte stores of dq | dk | dv rows ----
    for (int i = tid; i < S * 8; i += NT) {
        const int srow = i >> 3, ch = i & 7;
        const size_t m = (size_t)nb * S + srow;
        const f32x4 a = *(const f32x4*)(sdQ + srow * DQP + 8 * ch), b = *(const f32x4*)(sdQ + srow * DQP + 8 * ch + 4);
        const f32x4 cs0 = *(const f32x4*)(rope_cs + (size_t)srow * 64 + 8 * ch), cs1 = *(const f32x4*)(rope_cs + (size_t)srow * 64 + 8 * ch + 4);
        union { f16x4 h[2]; uint4 u; } o;
        o.h[0] = sat4(a[0] * cs0[0] + a[1] * cs0[1], a[1] * cs0[0] - a[0] * cs0[1], a[2] * cs0[2] + a[3] * cs0[3], a[3] * cs0[2] - a[2] * cs0[3], amax);
        o.h[1] = sat4(b[0] * cs1[0] + b[1] * cs1[1], b[1] * cs1[0] - b[0] * cs1[1], b[2] * cs1[2] + b[3] * cs1[3], b[3] * cs1[2] - b[2] * cs1[3], amax);
        *(uint4*)(dqkv + tiled_off((int)m, head * 64 + 8 * ch, 3 * D)) = o.u;
        *(uint4*)(dqkv + tiled_off((int)m, D + head * 64 + 8 * ch, 3 * D)) = *(const uint4*)(sK + srow * LP + 8 * ch);
        *(uint4*)(dqkv + tiled_off((int)m, 2 * D + head * 64 + 8 * ch, 3 * D)) = *(const uint4*)(sV + srow * LP + 8 * ch);
    }
    sat_report(amax, err_flag);
}

// ------------------------------------------------------------------------------------------------------------------------
// Temporal (causal) attention backward (model/attention.py:41-71): per (b, position p, head) a T x T lower-triangular problem,
// T <= 8.  16 lanes per item, 4 head features per lane, dot products by 4 xor-shuffles inside the 16-lane group.
// q fp16 [M][D] (m = (b T + t) P + p), kv cache [B][Tmax][P][2][D] (k with RoPE, v), dO fp16 row-major [M][D];
// output dqkv fp16 tile-major [M][3 D] with RoPE^T on dq / dk (position = frame index t).
// ------------------------------------------------------------------------------------------------------------------------
template <int TT>
__global__ __launch_bounds__(256) void attn_temporal_bwd_kernel(const f16* __restrict__ q, const f16* __restrict__ kv, const f16* __restrict__ dO, int B,
                                                                int P, int D, int Tmax, const float* __restrict__ rope_cs, f16* __restrict__ dqkv,
                                                                int* err_flag) {
    const int heads = D >> 6;
    const int gidx = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4), l = threadIdx.x & 15;
    const int items = B * P * heads;
    const bool valid = gidx < items;
    const int it = valid ? gidx : items - 1;
    const int head = it % heads, p = (it / heads) % P, b = it / (heads * P);
    const int col = head * 64 + 4 * l;
    float qv[TT][4], kk[TT][4], vv[TT][4], go[TT][4];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const size_t m = ((size_t)b * TT + t) * P + p;
        const f16x4 a = *(const f16x4*)(q + m * D + col), g = *(const f16x4*)(dO + m * D + col);
        const size_t slot = (((size_t)b * Tmax + t) * P + p) * 2 * D;
        const f16x4 kx = *(const f16x4*)(kv + slot + col), vx = *(const f16x4*)(kv + slot + D + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) { qv[t][e] = (float)a[e]; go[t][e] = (float)g[e]; kk[t][e] = (float)kx[e]; vv[t][e] = (float)vx[e]; }
    }
    auto dot16 = [&](const float (&a)[4], const float (&b2)[4]) {
        float s = (a[0] * b2[0] + a[1] * b2[1]) + (a[2] * b2[2] + a[3] * b2[3]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        return s;
    };
    float dq[TT][4], dk[TT][4], dv[TT][4];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) dq[t][e] = dk[t][e] = dv[t][e] = 0.f;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        float sc[TT], dp[TT], mx = -INFINITY;
#pragma unroll
        for (int s = 0; s <= t; ++s) {
            sc[s] = dot16(qv[t], kk[s]) * 0.125f;
            dp[s] = dot16(go[t], vv[s]);
            mx = fmaxf(mx, sc[s]);
        }
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s <= t; ++s) { sc[s] = __expf(sc[s] - mx); sum += sc[s]; }
        const float inv = 1.0f / sum;
        float dsum = 0.f;
#pragma unroll
        for (int s = 0; s <= t; ++s) { sc[s] *= inv; dsum += sc[s] * dp[s]; }
#pragma unroll
        for (int s = 0; s <= t; ++s) {
            const float ds = sc[s] * (dp[s] - dsum) * 0.125f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dq[t][e] += ds * kk[s][e];
                dk[s][e] += ds * qv[t][e];
                dv[s][e] += sc[s] * go[t][e];
            }
        }
    }
    float amax = 0.f;
    if (valid) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int m = (b * TT + t) * P + p;
            const f32x4 cs = *(const f32x4*)(rope_cs + (size_t)t * 64 + 4 * l);
            const float q0 = dq[t][0] * cs[0] + dq[t][1] * cs[1], q1 = dq[t][1] * cs[0] - dq[t][0] * cs[1];
            const float q2 = dq[t][2] * cs[2] + dq[t][3] * cs[3], q3 = dq[t][3] * cs[2] - dq[t][2] * cs[3];
            const float k0 = dk[t][0] * cs[0] + dk[t][1] * cs[1], k1 = dk[t][1] * cs[0] - dk[t][0] * cs[1];
            const float k2 = dk[t][2] * cs[2] + dk[t][3] * cs[3], k3 = dk[t][3] * cs[2] - dk[t][2] * cs[3];
            *(f16x4*)(dqkv + tiled_off(m, col, 3 * D)) = sat4(q0, q1, q2, q3, amax);
            *(f16x4*)(dqkv + tiled_off(m, D + col, 3 * D)) = sat4(k0, k1, k2, k3, amax);
            *(f16x4*)(dqkv + tiled_off(m, 2 * D + col, 3 * D)) = sat4(dv[t][0], dv[t][1], dv[t][2], dv[t][3], amax);
        }
    }
    sat_report(amax, err_flag);
}

// ------------------------------------------------------------------------------------------------------------------------
// fp32 conditioning path (tens of rows): SiLU, the two small GEMM forms of a Linear's backward, and the adaLN mega-projection.
// ------------------------------------------------------------------------------------------------------------------------
__global__ void silu_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int R, int C) {
    const size_t total = (size_t)R * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C), r = (int)(idx / C);
        const float v = x[(size_t)r * ldx + c];
        y[(size_t)r * ldy + c] = v / (1.0f + expf(-v));
    }
}
__global__ void silu_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx, float* __restrict__ dx, int lddx, int R, int C) {
    const size_t total = (size_t)R * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C), r = (int)(idx / C);
        const float v = x[(size_t)r * ldx + c], s = 1.0f / (1.0f + expf(-v));
        dx[(size_t)r * lddx + c] = dy[(size_t)r * lddy + c] * s * (1.0f + v * (1.0f - s));
    }
}
// dW[n][k] += sum_r dY[r][n] X[r][k]      (one thread per (8 rows n, column k); R is tens of rows: per r one coalesced X load,
// eight broadcast dY values, eight FMAs)
__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(const float* __restrict__ dY, int lddy, const float* __restrict__ X, int ldx, int R, int N, int K,
                                                          float* __restrict__ dW, int lddw) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int n0 = blockIdx.y * 8;
    if (k >= K) return;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool full = n0 + 8 <= N && (lddy & 3) == 0 && (((size_t)dY & 15) == 0) && (n0 & 3) == 0;
    for (int r = 0; r < R; ++r) {
        const float x = X[(size_t)r * ldx + k];
        const float* dy = dY + (size_t)r * lddy + n0;
        if (full) {
            const f32x4 d0 = *(const f32x4*)dy, d1 = *(const f32x4*)(dy + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] += d0[i] * x; a[4 + i] += d1[i] * x; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] += (n0 + i < N ? dy[i] : 0.f) * x;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (n0 + i < N) dW[(size_t)(n0 + i) * lddw + k] += a[i];
}
// The same on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: an fp32 fma chain): a wave owns 64 rows n x 64 columns k (4 x 4 tiles); per
// step of 4 conditioning rows r it loads 4 + 4 operand values per lane (64-byte row segments of dY and X, both L2-resident) for 16 MFMAs.
// N % 64 == 0, K % 64 == 0.  The VALU kernel above spent 78 us per half-block slice (6144 x 1024) on 3 load instructions per 8 FMAs;
// the read-modify-write of the 25 MB gradient slice is 13 us of that.
__global__ __launch_bounds__(256) void gemm_tn_f32_mfma_kernel(const float* __restrict__ dY, int lddy, const float* __restrict__ X, int ldx, int R, int N, int K,
                                                               float* __restrict__ dW, int lddw) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
    const int k0 = blockIdx.x * 64, n0 = (blockIdx.y * 4 + w) * 64;
    if (n0 >= N) return;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* dy = dY + n0 + li;
    const float* x = X + k0 + li;
    // the gradient tile this wave accumulates into is requested FIRST (its read used to start after the main loop), and the operand values of sixteen conditioning rows
    // are requested together (four dependent round trips per sixteen rows before): the launch is a handful of memory round trips, not FLOPs
    f32x4 old[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) old[a][b] = *(const f32x4*)(dW + (size_t)(n0 + 16 * a + li) * lddw + k0 + 4 * g + 16 * b);
    for (int r0 = 0; r0 < R; r0 += 16) {
        float av[4][4], bv[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r0 + 4 * q + g;
            const bool ok = r < R;
            const size_t rr = ok ? r : 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                av[q][t] = dy[rr * lddy + 16 * t];
                bv[q][t] = x[rr * ldx + 16 * t];
                if (!ok) av[q][t] = 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)     // rows in ascending order per output: the fma chain of the loop this replaces
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[q][b], av[q][a], acc[a][b], 0, 0, 0);
    }
    // X is the A operand: D[row = k 16 b + 4 g + e][col = n 16 a + li] — a lane holds four CONSECUTIVE k of one n: one 16-byte read-modify-write per tile
    // (with dY as the A operand it held four n of one k: sixteen 4-byte read-modify-writes per lane where there are four now; same products, same order)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float* o = dW + (size_t)(n0 + 16 * a + li) * lddw + k0 + 4 * g;
#pragma unroll
        for (int b = 0; b < 4; ++b) *(f32x4*)(o + 16 * b) = old[a][b] + acc[a][b];
    }
}
// dX[r][k] = sum_n dY[r][n] W[n][k]       (one thread per (r, k))
// Block = one wave = 64 columns k x up to 16 rows r (80 blocks for the 80 x 1024 product: a wider block leaves most of the chip idle); the rows of dY sit in LDS (broadcast reads), W[n][k] is loaded once per n for the block's rows.  One thread per (r, k)
// with its own pass over W was 370 us for the 80 x 1024 x 1024 product of the conditioning path's backward (a first rewrite that kept dY in global memory: 570 us).
template <int RB>
__global__ __launch_bounds__(64) void gemm_nn_f32_kernel(const float* __restrict__ dY, int lddy, const float* __restrict__ W, int ldw, int R, int N, int K,
                                                          float* __restrict__ dX, int lddx) {
    extern __shared__ __attribute__((aligned(16))) float sdy[];   // [RB][N]  (N % 4 == 0)
    const int k = blockIdx.x * 64 + threadIdx.x;
    const int r0 = blockIdx.y * RB;
    for (int idx = threadIdx.x; idx < RB * (N / 4); idx += 64) {
        const int i = idx / (N / 4), c = idx - i * (N / 4);
        const int r = r0 + i < R ? r0 + i : R - 1;              // (clamped rows compute a value nobody stores)
        *(f32x4*)(sdy + (size_t)i * N + 4 * c) = *(const f32x4*)(dY + (size_t)r * lddy + 4 * c);
    }
    __syncthreads();
    if (k >= K) return;
    float a[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) a[i] = 0.f;
    // sixteen rows of W at a time, the next sixteen requested before the current ones are used (N % 16 == 0, host-checked): with four loads per iteration and
    // nothing in flight across iterations the loop was a chain of 256 memory round trips (160 us)
    float w[16], wn[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) w[e] = W[(size_t)e * ldw + k];
    for (int n = 0; n < N; n += 16) {
        if (n + 16 < N) {
#pragma unroll
            for (int e = 0; e < 16; ++e) wn[e] = W[(size_t)(n + 16 + e) * ldw + k];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const f32x4 d = *(const f32x4*)(sdy + (size_t)i * N + n + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[i] += d[e] * w[4 * q + e];   // n ascending: the order of the kernel it replaces
            }
#pragma unroll
        for (int e = 0; e < 16; ++e) w[e] = wn[e];
    }
#pragma unroll
    for (int i = 0; i < RB; ++i)
        if (r0 + i < R) dX[(size_t)(r0 + i) * lddx + k] = a[i];
}
// The adaLN projection mod = SiLU(c) W_ada^T + b_ada with W_ada [MODW][D] (MODW ~ 2 x 10^5): dSc[r][n] = sum_k dmod[r][k] W_ada[k][n].
// Block = (256 features n) x (a chunk of KC rows k of W_ada), up to ADA_RB conditioning rows per pass in registers; W_ada is read once
// per pass (0.8 GB at full size); the partial sums of chunk c are STORED to part[c][r][n] and ada_reduce_kernel adds the chunks in a fixed order
// (round 4: this fallback added them with float atomics — the one place left where the step was not bit-reproducible).
constexpr int ADA_RB = 40;   // conditioning rows per pass (registers): the training batch of 16 x 5 rows takes two passes over W_ada
__global__ __launch_bounds__(256) void ada_bwd_dx_kernel(const float* __restrict__ dmod, int MODW, const float* __restrict__ W, int D, int R, int KC,
                                                         float* __restrict__ part) {
    __shared__ float sd[ADA_RB][256];
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int k0 = blockIdx.y * KC, k1 = min(MODW, k0 + KC);
    for (int rb = 0; rb < R; rb += ADA_RB) {
        float acc[ADA_RB];
#pragma unroll
        for (int i = 0; i < ADA_RB; ++i) acc[i] = 0.f;
        for (int kb = k0; kb < k1; kb += 256) {
            __syncthreads();
            for (int i = 0; i < ADA_RB; ++i) {
                const int k = kb + threadIdx.x;
                sd[i][threadIdx.x] = (rb + i < R && k < k1) ? dmod[(size_t)(rb + i) * MODW + k] : 0.f;
            }
            __syncthreads();
            const int kn = min(256, k1 - kb);
            if (n < D)
                for (int kk = 0; kk < kn; ++kk) {
                    const float w = W[(size_t)(kb + kk) * D + n];
#pragma unroll
                    for (int i = 0; i < ADA_RB; ++i) acc[i] += sd[i][kk] * w;
                }
        }
        if (n < D)
#pragma unroll
            for (int i = 0; i < ADA_RB; ++i)
                if (rb + i < R) part[((size_t)blockIdx.y * R + rb + i) * D + n] = acc[i];
    }
}
// dSc on the fp32 matrix cores.  One block of D / 64 waves per chunk of 1024 rows k of W_ada: W_ada is read ONCE (the VALU kernel
// above needs one pass per 40 conditioning rows and is FMA-bound: 2.2 ms per step at full size), every wave owns 64 columns n (4 tiles) x
// all R <= 80 rows (5 tiles); the dmod chunk goes through LDS 64 columns at a time (A operand: lane (li, g) reads row 16 t + li, column
// 4 s + g), W rows stream straight into registers (B operand: 64-byte row segments).  The partial sums of a chunk are stored, not
// added: ada_reduce_kernel sums the chunks in a fixed order (deterministic, no atomics).
constexpr int ADA_KC = 1024, ADA_RT = 5, ADA_PITCH = 68;
__global__ __launch_bounds__(1024) void ada_bwd_dx_mfma_kernel(const float* __restrict__ dmod, int MODW, const float* __restrict__ W, int D, int R,
                                                               float* __restrict__ part) {
    __shared__ float sd[16 * ADA_RT * ADA_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, g = lane >> 4, nthr = blockDim.x;
    const int n0 = w * 64;
    const int kc0 = blockIdx.x * ADA_KC, kc1 = min(MODW, kc0 + ADA_KC);
    f32x4 acc[ADA_RT][4];
#pragma unroll
    for (int t = 0; t < ADA_RT; ++t)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = kc0; kb < kc1; kb += 64) {
        __syncthreads();
        for (int i = tid; i < 16 * ADA_RT * 64; i += nthr) {
            const int r = i >> 6, kk = i & 63;
            sd[r * ADA_PITCH + kk] = (r < R && kb + kk < kc1) ? dmod[(size_t)r * MODW + kb + kk] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int s4 = 0; s4 < 16; ++s4) {
            int krow = kb + 4 * s4 + g;
            krow = krow < MODW ? krow : MODW - 1;           // rows past the chunk end meet zero dmod values
            const float* wp = W + (size_t)krow * D + n0 + li;
            float bv[4], av[ADA_RT];
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = wp[16 * b];
#pragma unroll
            for (int t = 0; t < ADA_RT; ++t) av[t] = sd[(16 * t + li) * ADA_PITCH + 4 * s4 + g];
#pragma unroll
            for (int t = 0; t < ADA_RT; ++t)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[t][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[b], acc[t][b], 0, 0, 0);
        }
    }
    // D[row = r 16 t + 4 g + e][col = n 16 b + li]
    float* o = part + (size_t)blockIdx.x * R * D;
#pragma unroll
    for (int t = 0; t < ADA_RT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 16 * t + 4 * g + e;
            if (r < R)
#pragma unroll
                for (int b = 0; b < 4; ++b) o[(size_t)r * D + n0 + 16 * b + li] = acc[t][b][e];
        }
}
__global__ void ada_reduce_kernel(const float* __restrict__ part, int nchunk, size_t n, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = 0.f;
    for (int c = 0; c < nchunk; ++c) a += part[(size_t)c * n + i];
    out[i] = a;
}

// ------------------------------------------------------------------------------------------------------------------------
// Optimizer (torch.optim.AdamW semantics, train_dit.py:232-238) with the loss scale and the clipping coefficient folded in.
//   ctl[0] = sum of squares of the SCALED gradients (sumsq_kernel, all parameters), ctl[1] = coefficient written by clip_coef_kernel:
//   inv_scale * min(1, max_norm / (norm + 1e-6)) (torch.nn.utils.clip_grad_norm_), or 0 when the norm is not finite (overflow:
//   the step is skipped and ctl[2] counts it).  Overflow = a non-finite norm OR the handle's error word carrying ERR_F16_SAT /
//   ERR_NONFINITE: every fp16 gradient / activation store saturates to +-65504 and raises that bit, so at a too-large loss scale the
//   norm itself stays finite — the bit is what says the gradients are clipped garbage.  The bits are cleared here (consumed).
//   Data-parallel: the bit is per rank, so the backward pass ends by turning it into +inf in one element of the rank's gradient arena
//   (overflow_publish_kernel, in the bucket that is all-reduced last): after the all-reduce EVERY rank's norm is non-finite and every rank skips —
//   weights, moments and the Adam step count cannot drift apart between replicas (round 3 tested the bit locally only: ADVICE r3).
//   ctl[4] = number of APPLIED steps (the Adam step count t: a skipped step does not advance it), ctl[5] / ctl[6] = the bias
//   corrections 1 - beta1^t / 1 - beta2^t of the step being applied, computed here on the device so that m / v and t stay in step.
// ------------------------------------------------------------------------------------------------------------------------
// (per-block partial sums, added up in block order by clip_coef_kernel: no float atomics, the norm — and with it the clip coefficient and every
// AdamW update — is bit-reproducible from run to run)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n, float* __restrict__ part) {
    // 16-byte loads, four of them in flight per thread, four independent sums (one 4-byte load per dependent add streamed the 2.4 GB arena at 4 TB/s)
    __shared__ float red[16];
    const f32x4* g4 = (const f32x4*)g;                  // (the arena is 16-byte aligned)
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * blockDim.x;
    f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const f32x4 v0 = g4[i], v1 = g4[i + stride], v2 = g4[i + 2 * stride], v3 = g4[i + 3 * stride];
        a0 = a0 + v0 * v0; a1 = a1 + v1 * v1; a2 = a2 + v2 * v2; a3 = a3 + v3 * v3;
    }
    for (; i < n4; i += stride) {
        const f32x4 v = g4[i];
        a0 = a0 + v * v;
    }
    const f32x4 q = (a0 + a1) + (a2 + a3);
    float a = (q[0] + q[1]) + (q[2] + q[3]);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t j = n4 << 2; j < n; ++j) a += g[j] * g[j];
    const float t = block_sum(a, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ __launch_bounds__(256) void clip_coef_kernel(float* ctl, const float* __restrict__ part, int nparts, float inv_scale, float max_norm, float beta1, float beta2,
                                                        int* err_flag) {
    __shared__ float red[16];
    float a = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) a += part[i];
    const float ssq = block_sum(a, red);
    if (threadIdx.x != 0) return;
    ctl[0] = ssq;
    const float norm = sqrtf(ssq) * inv_scale;
    bool overflow = !(norm == norm) || isinf(norm);
    if (err_flag) {
        const int bits = *err_flag & (ERR_F16_SAT | ERR_NONFINITE);
        if (bits) {
            overflow = true;
            atomicAnd(err_flag, ~bits);
        }
    }
    if (overflow) {
        ctl[1] = 0.f;
        ctl[2] += 1.f;
    } else {
        ctl[1] = inv_scale * (max_norm > 0.f ? fminf(1.0f, max_norm / (norm + 1e-6f)) : 1.0f);
        const float t = ctl[4] + 1.0f;
        ctl[4] = t;
        ctl[5] = 1.0f - powf(beta1, t);
        ctl[6] = 1.0f - powf(beta2, t);
    }
    ctl[3] = norm;
}
__global__ void adamw_kernel(float* __restrict__ p, int ldp, int R, int C, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             const float* __restrict__ ctl, float lr, float beta1, float beta2, float eps, float wd) {
    const float coef = ctl[1];
    if (coef == 0.f) return;               // overflow: skip the step
    const float bc1 = ctl[5], bc2 = ctl[6];
    const size_t total = (size_t)R * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const size_t r = idx / C;
        const float gr = g[idx] * coef;
        const float mm = beta1 * m[idx] + (1.0f - beta1) * gr;
        const float vv = beta2 * v[idx] + (1.0f - beta2) * gr * gr;
        m[idx] = mm;
        v[idx] = vv;
        float* pp = p + r * ldp + c;
        const float w = *pp * (1.0f - lr * wd);
        *pp = w - (lr / bc1) * mm / (sqrtf(vv) / sqrtf(bc2) + eps);
    }
}

// Multi-tensor AdamW (one launch for the whole model) fused with the refresh of the fp16 GEMM operands.  The host cuts every parameter
// into work items (api_train.hip gtav_dit_train_enable): a 64 x 64 tile of a GEMM weight, or a run of up to 4096 elements of an fp32
// parameter.  A block updates its item in fp32 and, for a GEMM weight, writes the tile straight into the tile-major fp16 W (16 bytes
// along c) and, through an LDS transpose, into the tile-major fp16 W^T (16 bytes along r): the separate convert passes (two more reads
// of the 0.8 GB of masters, 370 launches) and 300 per-parameter AdamW launches are gone.
__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamParam* __restrict__ params, const AdamItem* __restrict__ items, const float* __restrict__ ctl,
                                                          float lr, float beta1, float beta2, float eps, float wd) {
    const float coef = ctl[1];
    if (coef == 0.f) return;               // overflow: skip the step
    const float bc1 = ctl[5], bc2 = ctl[6];
    const AdamItem it = items[blockIdx.x];
    const AdamParam P = params[it.param];
    const float decay = 1.0f - lr * wd, step = lr / bc1, rs2 = 1.0f / sqrtf(bc2);
    auto upd = [&](size_t idx, float* pp) -> float {
        const float gr = P.g[idx] * coef;
        const float mm = beta1 * P.m[idx] + (1.0f - beta1) * gr;
        const float vv = beta2 * P.v[idx] + (1.0f - beta2) * gr * gr;
        P.m[idx] = mm;
        P.v[idx] = vv;
        const float w = *pp * decay - step * mm / (sqrtf(vv) * rs2 + eps);
        *pp = w;
        return w;
    };
    if (!P.w16) {                          // fp32 parameter (trained in place, possibly a strided view): a linear run of elements
        const size_t total = (size_t)P.R * P.C;
        const size_t end = (size_t)it.start + 4096 < total ? (size_t)it.start + 4096 : total;
        for (size_t idx = (size_t)it.start + threadIdx.x; idx < end; idx += 256) {
            const size_t r = idx / P.C;
            const int c = (int)(idx - r * P.C);
            upd(idx, P.p + r * P.ldp + c);
        }
        return;
    }
    __shared__ f16 t[64][72];
    const int tiles_c = (P.C + 63) / 64;
    const int r0 = (int)(it.start / tiles_c) * 64, c0 = (int)(it.start % tiles_c) * 64;
    if (r0 + 64 <= P.R && c0 + 64 <= P.C && (P.C & 3) == 0) {
        // interior tile (every tile of the DiT's GEMM weights): 16-byte accesses, and ALL of the thread's loads (4 float4 of each of g, m, v, w) before its first
        // store — a load behind a store waits for the store's round trip (vmcnt retires in order), three times per block in the element-wise loop below:
        // 5.41 -> 3.26 ms per step for the 608 M parameters (6 TB/s)
        f32x4 g4[4], m4[4], v4[4], w4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = threadIdx.x + 256 * k, r = q >> 4, c4 = (q & 15) * 4;
            const size_t idx = (size_t)(r0 + r) * P.C + c0 + c4;
            g4[k] = *(const f32x4*)(P.g + idx); m4[k] = *(const f32x4*)(P.m + idx); v4[k] = *(const f32x4*)(P.v + idx); w4[k] = *(const f32x4*)(P.p + idx);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = threadIdx.x + 256 * k, r = q >> 4, c4 = (q & 15) * 4;
            const size_t idx = (size_t)(r0 + r) * P.C + c0 + c4;
            f32x4 mm, vv, ww;
#pragma unroll
            for (int e = 0; e < 4; ++e) {   // (the same operations in the same order as upd())
                const float gr = g4[k][e] * coef;
                mm[e] = beta1 * m4[k][e] + (1.0f - beta1) * gr;
                vv[e] = beta2 * v4[k][e] + (1.0f - beta2) * gr * gr;
                ww[e] = w4[k][e] * decay - step * mm[e] / (sqrtf(vv[e]) * rs2 + eps);
                t[r][c4 + e] = (f16)__builtin_amdgcn_fmed3f(ww[e], -F16_MAX, F16_MAX);
            }
            *(f32x4*)(P.m + idx) = mm; *(f32x4*)(P.v + idx) = vv; *(f32x4*)(P.p + idx) = ww;
        }
    } else
    for (int q = threadIdx.x; q < 64 * 16; q += 256) {     // 64 rows x 16 float4
        const int r = q >> 4, c4 = (q & 15) * 4;
        f16 h4[4] = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        if (r0 + r < P.R) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = c0 + c4 + e;
                if (c < P.C) {
                    const size_t idx = (size_t)(r0 + r) * P.C + c;
                    float w = upd(idx, P.p + idx);
                    w = __builtin_amdgcn_fmed3f(w, -F16_MAX, F16_MAX);
                    h4[e] = (f16)w;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) t[r][c4 + e] = h4[e];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 512; q += 256) {
        const int a = q >> 3, ch = q & 7;
        // W: row r0 + a, columns c0 + 8 ch .. (rows / columns past the logical shape stay zero in both images)
        if (r0 + a < P.R && c0 + 8 * ch < P.Cp16) *(uint4*)(P.w16 + tiled_off(r0 + a, c0 + 8 * ch, P.Cp16)) = *(const uint4*)&t[a][8 * ch];
        if (P.wT && c0 + a < P.C && r0 + 8 * ch < P.RpT) {
            union { f16 h[8]; uint4 u; } o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o.h[i] = t[8 * ch + i][a];
            *(uint4*)(P.wT + tiled_off(c0 + a, r0 + 8 * ch, P.RpT)) = o.u;
        }
    }
}

static int grid_for(size_t n, int block = 256) { return (int)((n + block - 1) / block < 4096 ? (n + block - 1) / block : 4096); }

}  // namespace

int launch_transpose_tiled_f16(const f16* src, int R, int C, f16* dst, hipStream_t stream) {
    GTAV_REQUIRE(R > 0 && C > 0 && C % 64 == 0, "transpose: bad shape %d x %d", R, C);
    const int Rp = round_up(R, 64);
    hipLaunchKernelGGL(transpose_tiled_kernel, dim3(C / 64, Rp / 64), dim3(256), 0, stream, src, R, C, dst, Rp);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_convert_T_f16(const float* src, int lds, int R, int C, f16* dst, hipStream_t stream) {
    const int Cp = round_up(C, 128), Rp = round_up(R, 64);
    hipLaunchKernelGGL(convert_T_kernel, dim3(grid_for((size_t)Cp * Rp)), dim3(256), 0, stream, src, lds, R, C, dst, Cp, Rp);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_gelu_tiled(const f16* u, f16* h, size_t n, hipStream_t stream) {
    GTAV_REQUIRE(n % 8 == 0, "gelu: element count must be a multiple of 8");
    hipLaunchKernelGGL(gelu_tiled_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, u, h, n / 8);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_gelu_bwd_tiled(const f16* dh, const f16* u, f16* du, size_t n, int* err_flag, hipStream_t stream) {
    GTAV_REQUIRE(n % 8 == 0, "gelu_bwd: element count must be a multiple of 8");
    hipLaunchKernelGGL(gelu_bwd_tiled_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, dh, u, du, n / 8, err_flag);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
// launch_gelu_bwd_tiled + launch_colsum_tiled_f16(du -> db) in one pass: dh / u / du tile-major [round_up(M, 128)][N]; ws = colsum_workspace(M, N) floats
int gelu_bwd_colsum_splits(int M) { return cdiv(round_up(M, 128), 512); }
int launch_gelu_bwd_tiled_colsum(const f16* dh, const f16* u, f16* du, int M, int N, float* db, float* ws, int* err_flag, hipStream_t stream) {
    GTAV_REQUIRE(N % 64 == 0 && M > 0 && ws, "gelu_bwd_colsum: M=%d N=%d", M, N);
    const int Mp = round_up(M, 128), splits = gelu_bwd_colsum_splits(M);
    hipLaunchKernelGGL(gelu_bwd_colsum_kernel, dim3(N / 64, splits), dim3(256), 0, stream, dh, u, du, M, Mp, N, cdiv(Mp, splits), ws, err_flag);
    if (db) hipLaunchKernelGGL(colsum_reduce_kernel, dim3(cdiv(N, 32)), dim3(256), 0, stream, ws, splits, N, db);   // (db == nullptr: the caller reduces ws later, launch_colsum_reduce_multi)
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_ln_mod_bwd(const float* dxn, const float* x, const float* scale, int mod_stride, int rows_per_mod, int M, int D, float* dres, int accumulate,
                      float* stats, hipStream_t stream) {
    GTAV_REQUIRE(D % 4 == 0 && D <= 2048, "ln_mod_bwd: D=%d", D);
    hipLaunchKernelGGL(ln_mod_bwd_kernel, dim3(M), dim3(round_up(D / 4, 64)), 0, stream, dxn, x, scale, mod_stride, rows_per_mod, M, D, dres, accumulate, stats);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
// fused form of launch_ln_mod_bwd + launch_frame_reduce_ln (D = 1024 or 2048; part: ln_bwd_fused_workspace(frames, P, D) floats)
bool ln_bwd_fused_ok(int D) { return D == 1024 || D == 2048 || D == 512 || D == 256; }
size_t ln_bwd_fused_workspace(int frames, int P, int D) { return (size_t)frames * (size_t)cdiv(P, 16) * 2 * (size_t)D; }
int launch_ln_mod_bwd_fused(const float* dxn, const float* x, const float* scale, int mod_stride, int frames, int P, int D, float* dres, int accumulate,
                            float* dshift, float* dscale, float* part, hipStream_t stream) {
    GTAV_REQUIRE(ln_bwd_fused_ok(D) && part && frames > 0 && P > 0, "ln_mod_bwd_fused: D=%d", D);
    const dim3 grid(cdiv(P, 16), frames);
    if (D == 256) hipLaunchKernelGGL(ln_mod_bwd_fused_kernel<1>, grid, dim3(256), 0, stream, dxn, x, scale, mod_stride, P, dres, accumulate, part);
    else if (D == 512) hipLaunchKernelGGL(ln_mod_bwd_fused_kernel<2>, grid, dim3(256), 0, stream, dxn, x, scale, mod_stride, P, dres, accumulate, part);
    else if (D == 1024) hipLaunchKernelGGL(ln_mod_bwd_fused_kernel<4>, grid, dim3(256), 0, stream, dxn, x, scale, mod_stride, P, dres, accumulate, part);
    else hipLaunchKernelGGL(ln_mod_bwd_fused_kernel<8>, grid, dim3(256), 0, stream, dxn, x, scale, mod_stride, P, dres, accumulate, part);
    hipLaunchKernelGGL(frame_partials_reduce_kernel, dim3(cdiv(D, 256), frames), dim3(256), 0, stream, part, cdiv(P, 16), D, dshift, dscale, mod_stride);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_frame_reduce_ln(const float* dxn, const float* x, const float* stats, int frames, int P, int D, float* dshift, float* dscale, int mod_stride,
                           hipStream_t stream) {
    hipLaunchKernelGGL(frame_reduce_ln_kernel, dim3(cdiv(D, 64), frames), dim3(256), 0, stream, dxn, x, stats, frames, P, D, dshift, dscale, mod_stride);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_gate_bwd(const float* dres, const float* gate, int mod_stride, int rows_per_mod, int M, int D, f16* dy_tiled, int* err_flag, hipStream_t stream) {
    GTAV_REQUIRE(D % 64 == 0, "gate_bwd: D=%d", D);
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(grid_for((size_t)M * (D / 4))), dim3(256), 0, stream, dres, gate, mod_stride, rows_per_mod, M, D, dy_tiled, err_flag);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_frame_reduce_gate(const float* dres, const f16* y, int frames, int P, int D, float* dgate, int mod_stride, hipStream_t stream) {
    hipLaunchKernelGGL(frame_reduce_gate_kernel, dim3(cdiv(D, 64), frames), dim3(256), 0, stream, dres, y, frames, P, D, dgate, mod_stride);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
// fused form of launch_gate_bwd + launch_frame_reduce_gate + launch_colsum_tiled_f16(dy -> db): ws holds frames x D floats
int launch_gate_bwd_fused(const float* dres, const f16* y, const float* gate, int mod_stride, int frames, int P, int D, f16* dy_tiled, float* dgate, float* db,
                          float* ws, int* err_flag, hipStream_t stream) {
    GTAV_REQUIRE(D % 64 == 0 && frames > 0 && P > 0 && ws, "gate_bwd_fused: D=%d frames=%d P=%d", D, frames, P);
    hipLaunchKernelGGL(gate_bwd_fused_kernel, dim3(D / 64, frames), dim3(256), 0, stream, dres, y, gate, mod_stride, P, D, dy_tiled, dgate, ws, err_flag);
    if (db) hipLaunchKernelGGL(colsum_reduce_kernel, dim3(cdiv(D, 32)), dim3(256), 0, stream, ws, frames, D, db);   // (db == nullptr: reduced later, launch_colsum_reduce_multi)
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_colsum_reduce_multi(const float* const* ws, float* const* db, const int* splits, const int* N, int njobs, hipStream_t stream) {
    GTAV_REQUIRE(njobs >= 1 && njobs <= 4, "colsum_reduce_multi: %d jobs", njobs);
    ColsumJobs jobs = {};
    int nmax = 0;
    for (int j = 0; j < njobs; ++j) {
        jobs.ws[j] = ws[j]; jobs.db[j] = db[j]; jobs.splits[j] = splits[j]; jobs.N[j] = N[j];
        nmax = N[j] > nmax ? N[j] : nmax;
    }
    hipLaunchKernelGGL(colsum_reduce_multi_kernel, dim3(cdiv(nmax, 32), njobs), dim3(256), 0, stream, jobs);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
size_t colsum_workspace(int M, int N) { return (size_t)cdiv(M, 256) * (size_t)round_up(N, 256); }
int launch_colsum_tiled_f16(const f16* dy, int M, int N, float* db, float* ws, hipStream_t stream) {
    GTAV_REQUIRE(N % 64 == 0, "colsum: N=%d must be a multiple of 64", N);
    const int splits = cdiv(M, 512);
    GTAV_REQUIRE(splits == 1 || ws, "colsum: %d row splits need the partial-sum workspace", splits);
    hipLaunchKernelGGL(colsum_tiled_kernel, dim3(N / 64, splits), dim3(256), 0, stream, dy, M, N, db, cdiv(M, splits), splits > 1 ? ws : nullptr);
    if (splits > 1) hipLaunchKernelGGL(colsum_reduce_kernel, dim3(cdiv(N, 32)), dim3(256), 0, stream, ws, splits, N, db);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_colsum_f32(const float* a, int lda, int M, int N, float* db, float* ws, hipStream_t stream) {
    const int splits = cdiv(M, 256);
    GTAV_REQUIRE(splits == 1 || ws, "colsum: %d row splits need the partial-sum workspace", splits);
    hipLaunchKernelGGL(colsum_f32_kernel, dim3(cdiv(N, 256), splits), dim3(256), 0, stream, a, lda, M, N, db, cdiv(M, splits), splits > 1 ? ws : nullptr);
    if (splits > 1) hipLaunchKernelGGL(colsum_reduce_kernel, dim3(cdiv(N, 32)), dim3(256), 0, stream, ws, splits, N, db);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_to_tiled_f16(const float* a, int M, int D, f16* out, int* err_flag, hipStream_t stream) {
    GTAV_REQUIRE(D % 64 == 0, "to_tiled: D=%d", D);
    hipLaunchKernelGGL(to_tiled_f16_kernel, dim3(grid_for((size_t)M * (D / 4))), dim3(256), 0, stream, a, M, D, out, err_flag);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_mse_bwd_patch(const float* vpred, const float* vtarget, int B, int T, int C, int H, int W, int p, float scale, f16* dfo, int ldf, int* err_flag,
                         hipStream_t stream) {
    GTAV_REQUIRE(ldf % 64 == 0 && ldf >= C * p * p, "mse_bwd: ldf=%d", ldf);
    const size_t total = (size_t)B * T * (H / p) * (W / p) * ldf;
    hipLaunchKernelGGL(mse_bwd_patch_kernel, dim3(grid_for(total)), dim3(256), 0, stream, vpred, vtarget, B, T, C, H, W, p, scale, dfo, ldf, err_flag);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
static int g_attn_bwd_valu = GTAV_ENV_INT("GTAV_ATTN_BWD_VALU", 0);   // experiments build: 1 = the VALU kernel (A/B runs)
static int g_attn_bwd_dbg = GTAV_ENV_INT("GTAV_ATTN_BWD_DBG", 0);     // experiments build: timing variants of the MFMA kernel (wrong results)
int launch_attn_spatial_bwd(const f16* Q, const f16* K, const f16* Vt, const f16* dO, int NB, int heads, int S, int D, const float* rope_cs, f16* dqkv,
                            int* err_flag, hipStream_t stream) {
    GTAV_REQUIRE(S > 0 && S <= AB_MAXS && S % 16 == 0 && D == heads * 64, "attn_spatial_bwd: S=%d (<= %d, %% 16), D=%d", S, AB_MAXS, D);
    constexpr int NT = 512;
    const size_t lds_valu = (size_t)4 * S * 72 * 2 + (size_t)2 * (NT / 16) * (S + 4) * 4;
    const size_t lds_mfma = (size_t)4 * S * ABM_LP * 2 + (size_t)S * ABM_DQP * 4 + (size_t)2 * S * 4 + (size_t)ABM_NW * 256 * 2;
    static unsigned long long attr_devs = 0;
    int dev = 0;
    GTAV_CHECK_HIP(hipGetDevice(&dev));
    if (!(attr_devs >> (dev & 63) & 1)) {
        GTAV_CHECK_HIP(hipFuncSetAttribute((const void*)attn_spatial_bwd_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        GTAV_CHECK_HIP(hipFuncSetAttribute((const void*)attn_spatial_bwd_mfma_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#ifdef GTAV_EXPERIMENTS
        GTAV_CHECK_HIP(hipFuncSetAttribute((const void*)attn_spatial_bwd_mfma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        GTAV_CHECK_HIP(hipFuncSetAttribute((const void*)attn_spatial_bwd_mfma_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        GTAV_CHECK_HIP(hipFuncSetAttribute((const void*)attn_spatial_bwd_mfma_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#endif
        attr_devs |= 1ull << (dev & 63);
    }
    GTAV_REQUIRE(lds_valu <= 160 * 1024 && lds_mfma <= 160 * 1024, "attn_spatial_bwd: S=%d needs %zu bytes of LDS", S, lds_valu > lds_mfma ? lds_valu : lds_mfma);
    if (g_attn_bwd_valu)
        hipLaunchKernelGGL(attn_spatial_bwd_kernel<NT>, dim3(NB * heads), dim3(NT), lds_valu, stream, Q, K, Vt, dO, heads, S, D, rope_cs, dqkv, err_flag);
#ifdef GTAV_EXPERIMENTS
    else if (g_attn_bwd_dbg == 1)
        hipLaunchKernelGGL(attn_spatial_bwd_mfma_kernel<1>, dim3(NB * heads), dim3(64 * ABM_NW), lds_mfma, stream, Q, K, Vt, dO, heads, S, D, rope_cs, dqkv, err_flag);
    else if (g_attn_bwd_dbg == 3)
        hipLaunchKernelGGL(attn_spatial_bwd_mfma_kernel<3>, dim3(NB * heads), dim3(64 * ABM_NW), lds_mfma, stream, Q, K, Vt, dO, heads, S, D, rope_cs, dqkv, err_flag);
    else if (g_attn_bwd_dbg == 7)
        hipLaunchKernelGGL(attn_spatial_bwd_mfma_kernel<7>, dim3(NB * heads), dim3(64 * ABM_NW), lds_mfma, stream, Q, K, Vt, dO, heads, S, D, rope_cs, dqkv, err_flag);
#endif
    else
        hipLaunchKernelGGL(attn_spatial_bwd_mfma_kernel<0>, dim3(NB * heads), dim3(64 * ABM_NW), lds_mfma, stream, Q, K, Vt, dO, heads, S, D, rope_cs, dqkv, err_flag);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_attn_temporal_bwd(const f16* q, const f16* kv, const f16* dO, int B, int P, int D, int T, int Tmax, const float* rope_cs, f16* dqkv, int* err_flag,
                             hipStream_t stream) {
    GTAV_REQUIRE(T >= 1 && T <= 8 && T <= Tmax && D % 64 == 0, "attn_temporal_bwd: T=%d", T);
    const size_t threads = (size_t)B * P * (D / 64) * 16;
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
#define GTAV_TB(TT) case TT: hipLaunchKernelGGL(attn_temporal_bwd_kernel<TT>, grid, block, 0, stream, q, kv, dO, B, P, D, Tmax, rope_cs, dqkv, err_flag); break
    switch (T) { GTAV_TB(1); GTAV_TB(2); GTAV_TB(3); GTAV_TB(4); GTAV_TB(5); GTAV_TB(6); GTAV_TB(7); GTAV_TB(8); }
#undef GTAV_TB
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_silu(const float* x, int ldx, float* y, int ldy, int R, int C, hipStream_t stream) {
    hipLaunchKernelGGL(silu_kernel, dim3(grid_for((size_t)R * C)), dim3(256), 0, stream, x, ldx, y, ldy, R, C);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_silu_bwd(const float* dy, int lddy, const float* x, int ldx, float* dx, int lddx, int R, int C, hipStream_t stream) {
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(grid_for((size_t)R * C)), dim3(256), 0, stream, dy, lddy, x, ldx, dx, lddx, R, C);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_gemm_tn_f32(const float* dY, int lddy, const float* X, int ldx, int R, int N, int K, float* dW, int lddw, hipStream_t stream) {
    if (N % 64 == 0 && K % 64 == 0)
        hipLaunchKernelGGL(gemm_tn_f32_mfma_kernel, dim3(K / 64, cdiv(N, 256)), dim3(256), 0, stream, dY, lddy, X, ldx, R, N, K, dW, lddw);
    else
        hipLaunchKernelGGL(gemm_tn_f32_kernel, dim3(cdiv(K, 256), cdiv(N, 8)), dim3(256), 0, stream, dY, lddy, X, ldx, R, N, K, dW, lddw);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_gemm_nn_f32(const float* dY, int lddy, const float* W, int ldw, int R, int N, int K, float* dX, int lddx, hipStream_t stream) {
    GTAV_REQUIRE(N % 16 == 0 && lddy % 4 == 0 && (size_t)4 * N * sizeof(float) <= 64 * 1024, "gemm_nn_f32: N=%d (a multiple of 16; 4 rows of dY must fit 64 KiB of LDS)", N);
    if ((size_t)16 * N * sizeof(float) <= 64 * 1024)   // rows of dY per block: as many of 16 / 8 / 4 as fit 64 KiB of LDS
        hipLaunchKernelGGL(gemm_nn_f32_kernel<16>, dim3(cdiv(K, 64), cdiv(R, 16)), dim3(64), (size_t)16 * N * sizeof(float), stream, dY, lddy, W, ldw, R, N, K, dX, lddx);
    else if ((size_t)8 * N * sizeof(float) <= 64 * 1024)
        hipLaunchKernelGGL(gemm_nn_f32_kernel<8>, dim3(cdiv(K, 64), cdiv(R, 8)), dim3(64), (size_t)8 * N * sizeof(float), stream, dY, lddy, W, ldw, R, N, K, dX, lddx);
    else
        hipLaunchKernelGGL(gemm_nn_f32_kernel<4>, dim3(cdiv(K, 64), cdiv(R, 4)), dim3(64), (size_t)4 * N * sizeof(float), stream, dY, lddy, W, ldw, R, N, K, dX, lddx);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
size_t ada_bwd_dx_workspace(int MODW, int D, int R) { return (size_t)cdiv(MODW, ADA_KC) * R * D; }
int launch_ada_bwd_dx(const float* dmod, int MODW, const float* W, int D, int R, float* dSc, float* part, hipStream_t stream) {
    if (part && D % 64 == 0 && D <= 1024 && R <= 16 * ADA_RT) {
        const int nchunk = cdiv(MODW, ADA_KC);
        hipLaunchKernelGGL(ada_bwd_dx_mfma_kernel, dim3(nchunk), dim3(D), 0, stream, dmod, MODW, W, D, R, part);
        GTAV_CHECK_HIP(hipGetLastError());
        const size_t n = (size_t)R * D;
        hipLaunchKernelGGL(ada_reduce_kernel, dim3((unsigned)cdiv((long long)n, 256)), dim3(256), 0, stream, part, nchunk, n, dSc);
    } else {
        // shapes the matrix-core kernel does not take (hidden % 64 != 0, more than 80 conditioning rows): the VALU kernel, same fixed-order reduction
        GTAV_REQUIRE(part, "ada_bwd_dx: needs its partial-sum workspace (ada_bwd_dx_workspace)");
        const int KC = 2048, nchunk = cdiv(MODW, KC);    // (<= cdiv(MODW, ADA_KC) chunks: the workspace covers it)
        hipLaunchKernelGGL(ada_bwd_dx_kernel, dim3(cdiv(D, 256), nchunk), dim3(256), 0, stream, dmod, MODW, W, D, R, KC, part);
        GTAV_CHECK_HIP(hipGetLastError());
        const size_t n = (size_t)R * D;
        hipLaunchKernelGGL(ada_reduce_kernel, dim3((unsigned)cdiv((long long)n, 256)), dim3(256), 0, stream, part, nchunk, n, dSc);
    }
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int sumsq_parts(size_t n) { return grid_for(n, 256 * 8); }
int launch_sumsq(const float* g, size_t n, float* part, hipStream_t stream) {
    hipLaunchKernelGGL(sumsq_kernel, dim3(sumsq_parts(n)), dim3(256), 0, stream, g, n, part);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
__global__ void overflow_publish_kernel(const int* __restrict__ err_flag, float* __restrict__ g) {
    if (*err_flag & (ERR_F16_SAT | ERR_NONFINITE)) *g = __builtin_inff();
}
__global__ void err_clear_kernel(int* err_flag, int bits) { atomicAnd(err_flag, ~bits); }
int launch_err_clear(int* err_flag, int bits, hipStream_t stream) {
    if (!err_flag) return 0;
    hipLaunchKernelGGL(err_clear_kernel, dim3(1), dim3(1), 0, stream, err_flag, bits);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_overflow_publish(const int* err_flag, float* g, hipStream_t stream) {
    if (!err_flag || !g) return 0;
    hipLaunchKernelGGL(overflow_publish_kernel, dim3(1), dim3(1), 0, stream, err_flag, g);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_clip_coef(float* ctl, const float* part, int nparts, float inv_scale, float max_norm, float beta1, float beta2, int* err_flag, hipStream_t stream) {
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, stream, ctl, part, nparts, inv_scale, max_norm, beta1, beta2, err_flag);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_adamw_multi(const AdamParam* params, const AdamItem* items, int n_items, const float* ctl, float lr, float beta1, float beta2, float eps, float wd,
                       hipStream_t stream) {
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(n_items), dim3(256), 0, stream, params, items, ctl, lr, beta1, beta2, eps, wd);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_adamw(float* p, int ldp, int R, int C, const float* g, float* m, float* v, const float* ctl, float lr, float beta1, float beta2, float eps,
                 float wd, hipStream_t stream) {
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for((size_t)R * C)), dim3(256), 0, stream, p, ldp, R, C, g, m, v, ctl, lr, beta1, beta2, eps, wd);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace gtav
