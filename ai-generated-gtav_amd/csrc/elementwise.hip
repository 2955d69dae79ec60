// Memory-bound kernels around the GEMMs: LayerNorm+modulate, patch gather/scatter, conversions,
// conditioning inputs, the DDIM update and the training-side noising / v-target / MSE.
// All are HBM-bound; every global access is 8-16 B per lane, rows are walked by whole waves.
#include "ops.h"

namespace gtav {

namespace {

// ------------------------------------------------------------------------------------------
// LayerNorm (eps 1e-6) over D, one wave per row, row kept in registers (D <= 2048, D % 4 == 0).
// MODE 0: adaLN modulate  y = xhat * (1 + (scale + 1e-6)) + shift     (model/dit.py:19-27)
// MODE 1: affine          y = xhat * gamma + beta                     (nn.LayerNorm, model/vae.py:174)
// ------------------------------------------------------------------------------------------
template <int MODE, int NV>
__global__ __launch_bounds__(256) void ln_kernel(const float* __restrict__ x, int ldx, f16* __restrict__ out, int ldo,
                                                 int M, int D, const float* __restrict__ p0, const float* __restrict__ p1,
                                                 int mod_stride, const int* __restrict__ rows, int rows_per_mod) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* xr = x + (size_t)m * ldx;
    f32x4 v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < D) {
            v[i] = *(const f32x4*)(xr + c);
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                sq += d * d;
            }
        }
    }
    const float var = wave_sum(sq) / (float)D;
    const float rstd = 1.0f / sqrtf(var + 1e-6f);
    const float *a, *b;  // MODE 0: a = scale row, b = shift row; MODE 1: a = gamma, b = beta
    if (MODE == 0) {
        int row = m / rows_per_mod;
        if (rows) row = rows[row];
        a = p1 + (size_t)row * mod_stride;
        b = p0 + (size_t)row * mod_stride;
    } else {
        a = p0;
        b = p1;
    }
    f16* orow = out + (size_t)m * ldo;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        if (c < D) {
            const f32x4 av = *(const f32x4*)(a + c);
            const f32x4 bv = *(const f32x4*)(b + c);
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (v[i][e] - mean) * rstd;
                float y;
                if (MODE == 0) {
                    const float sc = av[e] + 1e-6f;
                    y = xh * (1.0f + sc) + bv[e];
                } else {
                    y = xh * av[e] + bv[e];
                }
                o[e] = (f16)y;
            }
            *(f16x4*)(orow + c) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------
__global__ void patchify_kernel(const float* __restrict__ img, const int* __restrict__ frame_index, int NB, int C, int H,
                                int W, int p, f16* __restrict__ out, int ldo, float a, float b) {
    const int gh = H / p, gw = W / p, Kp = C * p * p;
    const size_t total = (size_t)NB * gh * gw * ldo;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx % ldo);
        const size_t m = idx / ldo;
        float val = 0.f;
        if (k < Kp) {
            const int pw = k % p, ph = (k / p) % p, c = k / (p * p);
            const int x = (int)(m % gw), y = (int)((m / gw) % gh), nb = (int)(m / ((size_t)gw * gh));
            const int f = frame_index ? frame_index[nb] : nb;
            val = a * img[(((size_t)f * C + c) * H + (y * p + ph)) * W + (x * p + pw)] + b;
        }
        out[idx] = (f16)val;
    }
}

__global__ void unpatchify_kernel(const float* __restrict__ y, int ldy, float* __restrict__ img, int NB, int C, int H, int W,
                                  int p, int order, float a, float b) {
    const int gh = H / p, gw = W / p;
    const size_t total = (size_t)NB * C * H * W;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(idx % W), yy = (int)((idx / W) % H), c = (int)((idx / ((size_t)W * H)) % C);
        const int nb = (int)(idx / ((size_t)W * H * C));
        const int ph = yy % p, pw = xx % p;
        const size_t m = ((size_t)nb * gh + yy / p) * gw + xx / p;
        const int f = order == 0 ? (ph * p + pw) * C + c : (c * p + ph) * p + pw;
        img[idx] = a * y[m * ldy + f] + b;
    }
}

__global__ void convert_pad_f16_kernel(const float* __restrict__ src, int lds, int R, int C, f16* __restrict__ dst, int Rp,
                                       int Cp, float scale) {
    const size_t total = (size_t)Rp * Cp;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % Cp);
        const size_t r = idx / Cp;
        float v = 0.f;
        if (r < (size_t)R && c < C) v = src[r * lds + c] * scale;
        dst[idx] = (f16)v;
    }
}

__global__ void copy_f32_kernel(const float* __restrict__ src, int lds, int R, int C, float* __restrict__ dst, int ldd, int c0) {
    const size_t total = (size_t)R * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const size_t r = idx / C;
        dst[r * ldd + c0 + c] = src[r * lds + c];
    }
}

__global__ void fill_f32_kernel(float* dst, size_t n, float v) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) dst[idx] = v;
}

__global__ void add_f32_kernel(const float* a, const float* b, float* out, size_t n) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x)
        out[idx] = a[idx] + b[idx];
}

__global__ void cond_inputs_kernel(const int64_t* __restrict__ t64, int rows, int Tq, int t_ctx, int t_cur,
                                   const float* __restrict__ sincos, float* __restrict__ E, const float* __restrict__ actions,
                                   long long act_outer, long long act_inner, int A, float* __restrict__ HC, int ldhc, int D,
                                   int Apad, int* err_flag) {
    const int r = blockIdx.x;
    if (r >= rows) return;
    const int ro = r / Tq, ri = r - ro * Tq;
    long long t = t64 ? (long long)t64[r] : (long long)(ri == Tq - 1 ? t_cur : t_ctx);
    if (t < 0 || t > 999) {
        if (threadIdx.x == 0 && err_flag) atomicOr(err_flag, 1);
        t = t < 0 ? 0 : 999;
    }
    for (int j = threadIdx.x; j < 256; j += blockDim.x) E[(size_t)r * 256 + j] = sincos[(size_t)t * 256 + j];
    const float* arow = actions ? actions + ro * act_outer + ri * act_inner : nullptr;
    for (int j = threadIdx.x; j < Apad; j += blockDim.x) HC[(size_t)r * ldhc + D + j] = (arow && j < A) ? arow[j] : 0.f;
}

__global__ void ddim_update_kernel(const float* __restrict__ x, size_t x_stride, const float* __restrict__ v, size_t v_stride,
                                   float* __restrict__ out, size_t out_stride, int n, const float* __restrict__ alpha_t,
                                   const float* __restrict__ alpha_next, float at_s, float an_s, int is_final) {
    const int b = blockIdx.y;
    const float at = alpha_t ? alpha_t[b] : at_s;
    const float an = alpha_t ? (alpha_next ? alpha_next[b] : 1.f) : an_s;
    const float s_at = sqrtf(at), s_1at = sqrtf(1.0f - at);
    const float s_rat = sqrtf(1.0f / at), s_den = sqrtf(1.0f / at - 1.0f);
    const float s_an = sqrtf(an), s_1an = sqrtf(1.0f - an);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float xc = x[b * x_stride + i], vp = v[b * v_stride + i];
        const float x0 = s_at * xc - s_1at * vp;
        float r = x0;
        if (!is_final) {
            const float eps = (s_rat * xc - x0) / s_den;
            r = s_an * x0 + s_1an * eps;
        }
        out[b * out_stride + i] = r;
    }
}

__global__ void add_noise_kernel(const float* __restrict__ x, const float* __restrict__ noise, const float* __restrict__ alpha,
                                 float* __restrict__ out, int n, float clamp_abs) {
    const int r = blockIdx.y;
    const float a = alpha[r];
    const float sa = sqrtf(a), s1 = sqrtf(1.0f - a);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const size_t o = (size_t)r * n + i;
        const float z = fminf(fmaxf(noise[o], -clamp_abs), clamp_abs);
        out[o] = x[o] * sa + s1 * z;
    }
}

__global__ void vtarget_kernel(const float* __restrict__ x, const float* __restrict__ noise, const float* __restrict__ alpha,
                               float* __restrict__ vt, int n, float clamp_abs) {
    const int r = blockIdx.y;
    const float a = alpha[r];
    const float sa = sqrtf(a), s1 = sqrtf(1.0f - a);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const size_t o = (size_t)r * n + i;
        const float z = fminf(fmaxf(noise[o], -clamp_abs), clamp_abs);
        vt[o] = sa * z - s1 * x[o];
    }
}

// deterministic two-stage mean of squared differences
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ a, size_t a_stride, const float* __restrict__ b,
                                                          size_t b_stride, int n, float* __restrict__ partial) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float d = a[r * a_stride + i] - b[r * b_stride + i];
        s += d * d;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[r] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void mse_final_kernel(const float* partial, int rows, float inv_count, float* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < rows; ++i) s += partial[i];
        *out = s * inv_count;
    }
}

__global__ void unpad_f16_kernel(const f16* __restrict__ src, int lds, int R, int C, float* __restrict__ dst) {
    const size_t total = (size_t)R * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const size_t r = idx / C;
        dst[idx] = (float)src[r * lds + c];
    }
}
__global__ void copy_rows_kernel(const float* __restrict__ src, size_t ss, float* __restrict__ dst, size_t ds, size_t n) {
    const int r = blockIdx.y;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[r * ds + i] = src[r * ss + i];
}
__global__ void clamp_cols_kernel(float* buf, int M, int ld, int c0, int c1, float lo, float hi) {
    const int w = c1 - c0;
    const size_t total = (size_t)M * w;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t m = idx / w;
        const int c = c0 + (int)(idx % w);
        const float v = buf[m * ld + c];
        buf[m * ld + c] = fminf(fmaxf(v, lo), hi);
    }
}
__global__ void frame_index_kernel(int* idx, int B, int Tq, int F, int first) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * Tq) idx[i] = (i / Tq) * F + first + (i % Tq);
}
// (N,3,H,W) f32 -> (N,H,W,3) u8 = clamp(img*255, 0, 255) truncated (torch .byte())
__global__ void frames_to_u8_kernel(const float* __restrict__ img, uint8_t* __restrict__ out, int N, int H, int W) {
    const size_t total = (size_t)N * H * W * 3;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % 3);
        const size_t pix = idx / 3;
        const size_t n = pix / ((size_t)H * W), yx = pix % ((size_t)H * W);
        const float v = fminf(fmaxf(img[(n * 3 + c) * (size_t)H * W + yx] * 255.0f, 0.0f), 255.0f);
        out[idx] = (uint8_t)v;
    }
}
// moments (N, hw, mom_ch) -> latents (N, latent, hw) = scale * moments[..., :latent]
__global__ void moments_to_latents_kernel(const float* __restrict__ mom, float* __restrict__ lat, int N, int hw, int latent,
                                          int mom_ch, float scale) {
    const size_t total = (size_t)N * latent * hw;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int s = (int)(idx % hw);
        const int c = (int)((idx / hw) % latent);
        const size_t n = idx / ((size_t)hw * latent);
        lat[idx] = mom[(n * hw + s) * mom_ch + c] * scale;
    }
}
// latents (N, latent, hw) -> tokens (N, hw, latent)
__global__ void latents_to_tokens_kernel(const float* __restrict__ lat, float* __restrict__ z, int N, int hw, int latent) {
    const size_t total = (size_t)N * latent * hw;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % latent);
        const int s = (int)((idx / latent) % hw);
        const size_t n = idx / ((size_t)hw * latent);
        z[idx] = lat[(n * latent + c) * hw + s];
    }
}

inline int grid_for(size_t total, int block = 256) {
    size_t g = (total + block - 1) / block;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

int launch_ln_modulate(const float* x, int ldx, f16* out, int ldo, int M, int D, const float* shift, const float* scale,
                       int mod_stride, const int* rows, int rows_per_mod, hipStream_t stream) {
    GTAV_REQUIRE(D % 4 == 0 && D <= 2048 && rows_per_mod > 0, "ln_modulate: D=%d must be %%4 and <= 2048", D);
#define LN_LAUNCH(NV)                                                                                            \
    hipLaunchKernelGGL((ln_kernel<0, NV>), dim3(cdiv(M, 4)), dim3(256), 0, stream, x, ldx, out, ldo, M, D, shift, scale, \
                       mod_stride, rows, rows_per_mod)
    if (D <= 256) LN_LAUNCH(1);
    else if (D <= 512) LN_LAUNCH(2);
    else if (D <= 1024) LN_LAUNCH(4);
    else LN_LAUNCH(8);
#undef LN_LAUNCH
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_ln_affine(const float* x, int ldx, f16* out, int ldo, int M, int D, const float* gamma, const float* beta,
                     hipStream_t stream) {
    GTAV_REQUIRE(D % 4 == 0 && D <= 2048, "ln_affine: D=%d must be %%4 and <= 2048", D);
#define LN_LAUNCH(NV)                                                                                           \
    hipLaunchKernelGGL((ln_kernel<1, NV>), dim3(cdiv(M, 4)), dim3(256), 0, stream, x, ldx, out, ldo, M, D, gamma, beta, 0, \
                       (const int*)nullptr, 1)
    if (D <= 256) LN_LAUNCH(1);
    else if (D <= 512) LN_LAUNCH(2);
    else if (D <= 1024) LN_LAUNCH(4);
    else LN_LAUNCH(8);
#undef LN_LAUNCH
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_patchify(const float* img, const int* frame_index, int NB, int C, int H, int W, int p, f16* out, int ldo,
                    float a, float b, hipStream_t stream) {
    GTAV_REQUIRE(H % p == 0 && W % p == 0 && ldo >= C * p * p, "patchify: bad geometry");
    const size_t total = (size_t)NB * (H / p) * (W / p) * ldo;
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total)), dim3(256), 0, stream, img, frame_index, NB, C, H, W, p, out,
                       ldo, a, b);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_unpatchify(const float* y, int ldy, float* img, int NB, int C, int H, int W, int p, int order, float a,
                      float b, hipStream_t stream) {
    GTAV_REQUIRE(H % p == 0 && W % p == 0 && ldy >= C * p * p, "unpatchify: bad geometry");
    const size_t total = (size_t)NB * C * H * W;
    hipLaunchKernelGGL(unpatchify_kernel, dim3(grid_for(total)), dim3(256), 0, stream, y, ldy, img, NB, C, H, W, p, order,
                       a, b);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_convert_pad_f16(const float* src, int lds, int R, int C, f16* dst, int Rp, int Cp, float scale, hipStream_t stream) {
    GTAV_REQUIRE(Rp >= R && Cp >= C, "convert_pad: padded shape smaller than source");
    hipLaunchKernelGGL(convert_pad_f16_kernel, dim3(grid_for((size_t)Rp * Cp)), dim3(256), 0, stream, src, lds, R, C, dst,
                       Rp, Cp, scale);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_copy_f32(const float* src, int lds, int R, int C, float* dst, int ldd, int c0, hipStream_t stream) {
    hipLaunchKernelGGL(copy_f32_kernel, dim3(grid_for((size_t)R * C)), dim3(256), 0, stream, src, lds, R, C, dst, ldd, c0);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_fill_f32(float* dst, size_t n, float v, hipStream_t stream) {
    hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(n)), dim3(256), 0, stream, dst, n, v);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_add_f32(const float* a, const float* b, float* out, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid_for(n)), dim3(256), 0, stream, a, b, out, n);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_cond_inputs(const int64_t* t64, int rows, int Tq, int t_ctx, int t_cur, const float* sincos, float* E,
                       const float* actions, int64_t act_outer, int64_t act_inner, int A, float* HC, int ldhc, int D, int Apad,
                       int* err_flag, hipStream_t stream) {
    GTAV_REQUIRE(rows > 0 && Tq > 0, "cond_inputs: bad rows/Tq");
    hipLaunchKernelGGL(cond_inputs_kernel, dim3(rows), dim3(256), 0, stream, t64, rows, Tq, t_ctx, t_cur, sincos, E, actions,
                       (long long)act_outer, (long long)act_inner, A, HC, ldhc, D, Apad, err_flag);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_ddim_update(const float* x, size_t x_stride, const float* v, size_t v_stride, float* out, size_t out_stride,
                       int B, int n, const float* alpha_t, const float* alpha_next, float alpha_t_s, float alpha_next_s,
                       int is_final, hipStream_t stream) {
    hipLaunchKernelGGL(ddim_update_kernel, dim3(cdiv(n, 256), B), dim3(256), 0, stream, x, x_stride, v, v_stride, out,
                       out_stride, n, alpha_t, alpha_next, alpha_t_s, alpha_next_s, is_final);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_add_noise(const float* x, const float* noise, const float* alpha, float* out, int rows, int n, float clamp_abs,
                     hipStream_t stream) {
    hipLaunchKernelGGL(add_noise_kernel, dim3(cdiv(n, 256), rows), dim3(256), 0, stream, x, noise, alpha, out, n, clamp_abs);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_vtarget(const float* x, const float* noise, const float* alpha, float* vt, int rows, int n, float clamp_abs,
                   hipStream_t stream) {
    hipLaunchKernelGGL(vtarget_kernel, dim3(cdiv(n, 256), rows), dim3(256), 0, stream, x, noise, alpha, vt, n, clamp_abs);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_mse(const float* a, size_t a_stride, const float* b, size_t b_stride, int rows, int n, float* out_scalar,
               hipStream_t stream) {
    // out_scalar[0] = mean; out_scalar[1 .. rows] is scratch for the per-row partial sums
    hipLaunchKernelGGL(mse_partial_kernel, dim3(rows), dim3(256), 0, stream, a, a_stride, b, b_stride, n, out_scalar + 1);
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, stream, out_scalar + 1, rows, 1.0f / ((float)rows * (float)n),
                       out_scalar);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_unpad_f16_to_f32(const f16* src, int lds, int R, int C, float* dst, hipStream_t stream) {
    hipLaunchKernelGGL(unpad_f16_kernel, dim3(grid_for((size_t)R * C)), dim3(256), 0, stream, src, lds, R, C, dst);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_copy_f32_strided(const float* src, int lds, int R, int C, float* dst, int ldd, hipStream_t stream) {
    return launch_copy_f32(src, lds, R, C, dst, ldd, 0, stream);
}
int launch_copy_rows_f32(const float* src, size_t src_stride, float* dst, size_t dst_stride, int rows, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(n), rows), dim3(256), 0, stream, src, src_stride, dst, dst_stride, n);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_clamp_cols(float* buf, int M, int ld, int c0, int c1, float lo, float hi, hipStream_t stream) {
    hipLaunchKernelGGL(clamp_cols_kernel, dim3(grid_for((size_t)M * (c1 - c0))), dim3(256), 0, stream, buf, M, ld, c0, c1, lo, hi);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_frame_index(int* idx, int B, int Tq, int F, int first, hipStream_t stream) {
    hipLaunchKernelGGL(frame_index_kernel, dim3(cdiv(B * Tq, 256)), dim3(256), 0, stream, idx, B, Tq, F, first);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_frames_to_u8(const float* img, uint8_t* out, int N, int H, int W, hipStream_t stream) {
    hipLaunchKernelGGL(frames_to_u8_kernel, dim3(grid_for((size_t)N * H * W * 3)), dim3(256), 0, stream, img, out, N, H, W);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_moments_to_latents(const float* mom, float* lat, int N, int hw, int latent, int mom_ch, float scale, hipStream_t stream) {
    hipLaunchKernelGGL(moments_to_latents_kernel, dim3(grid_for((size_t)N * hw * latent)), dim3(256), 0, stream, mom, lat, N, hw,
                       latent, mom_ch, scale);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_latents_to_tokens(const float* lat, float* z, int N, int hw, int latent, hipStream_t stream) {
    hipLaunchKernelGGL(latents_to_tokens_kernel, dim3(grid_for((size_t)N * hw * latent)), dim3(256), 0, stream, lat, z, N, hw, latent);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace gtav
