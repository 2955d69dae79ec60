// Memory-bound kernels around the GEMMs: LayerNorm+modulate, patch gather/scatter, conversions,
// conditioning inputs, the DDIM update and the training-side noising / v-target / MSE.
// All are HBM-bound; every global access is 8-16 B per lane, rows are walked by whole waves.
#include "ops.h"

#include <cstdlib>
#include <type_traits>

#include <cstring>

namespace gtav {

namespace {

// ------------------------------------------------------------------------------------------
// LayerNorm (eps 1e-6) over D, fp32 statistics, output fp16 tile-major (GEMM A-operand).
// MODE 0: adaLN modulate  y = xhat * (1 + (scale + 1e-6)) + shift     (model/dit.py:19-27)
// MODE 1: affine          y = xhat * gamma + beta                     (nn.LayerNorm, model/vae.py:174)
//
// ONE WAVE PER ROW (4 rows per 256-thread block) for the thousands of rows of the batched paths without a pending update (B = 8 window 5 760 rows,
// batched VAE 23 040 / 46 080): the row lives in registers (NV float4 per lane, D = 256 NV), the statistics are the same shifted one-pass sums as in
// ln_row_block_kernel below (shift K = the row's first element, here lane 0's register: no extra load) reduced by DPP moves only — no LDS, no barrier.
// Why not one block per row there: at these sizes that kernel is VALU-bound, not memory-bound — every thread of its four waves repeats the row-uniform
// tail (three IEEE divisions, a square root, the LDS sum: ~150 instructions) for FOUR elements, 46 080 rows x 256 threads x 150 instructions = 51 us of the
// chip's vector issue where a copy kernel moves the same 283 MB in 42 us.  A wave per row amortises the tail over 16 elements per lane and keeps
// 4 KiB per wave in flight (tools/ln_stream_probe.hip, profiles/round5/layernorm_large_m.txt: 68 -> 46 us at 46 080 rows).
// The sums are added in another order than in the row-block kernel: the two kernels agree to rounding, not to the bit.
// ------------------------------------------------------------------------------------------
// (With a pending update the same layout was measured too — slabs, gate and write-back per lane: 5.1 -> 7.5 us per launch at 720 rows in the batch-1
// forward, 22.4 -> 20-21 us at 5 760 rows, 32.0 -> 31.6 at 11 520: not kept, those launches stay on the row-block kernel.)
template <int MODE, int NV>
__global__ __launch_bounds__(256) void ln_wave_row_kernel(const float* __restrict__ x, int ldx, f16* __restrict__ out, int M, int D,
                                                          const float* __restrict__ p0, const float* __restrict__ p1, int mod_stride,
                                                          const int* __restrict__ rows, int rows_per_mod, int flags, int* err_flag) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* xr = x + (size_t)m * ldx;
    f32x4 v[NV], av[NV], bv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *(const f32x4*)(xr + i * 256 + lane * 4);
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
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        av[i] = *(const f32x4*)(a + i * 256 + lane * 4);
        bv[i] = *(const f32x4*)(b + i * 256 + lane * 4);
    }
    const float kshift = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v[0][0])));
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float a0 = v[i][0] - kshift, a1 = v[i][1] - kshift, a2 = v[i][2] - kshift, a3 = v[i][3] - kshift;
        s1 += (a0 + a1) + (a2 + a3);
        s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    const float t1 = wave_sum_dpp(s1), t2 = wave_sum_dpp(s2);
    const float m1 = t1 / (float)D, mean = kshift + m1;
    const float var = fmaxf(t2 / (float)D - m1 * m1, 0.f);
    const float rstd = 1.0f / sqrtf(var + 1e-6f);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float yv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (v[i][e] - mean) * rstd;
            if (MODE == 0) {
                const float sc = av[i][e] + 1e-6f;
                yv[e] = xh * (1.0f + sc) + bv[i][e];
            } else {
                yv[e] = xh * av[i][e] + bv[i][e];
            }
        }
        store_f16x4_paired<1>(out + tiled_off(m, i * 256 + lane * 4, D), sat4(yv[0], yv[1], yv[2], yv[3], amax), lane, flags & 2);
    }
    sat_report(amax, err_flag);
}

// ONE BLOCK PER ROW (D/4 threads, one float4 each), statistics through LDS: every launch with a pending update (split-K slabs, gate, residual
// write-back) and every launch of a few hundred rows, where a wave per row leaves most CUs idle (720 rows: 8.5 us vs 14.7) and the launch is
// latency-bound.  Optional deferred residual update first (LnPending).
#ifdef GTAV_EXPERIMENTS   // timing experiments (WRONG results): GTAV_LN_FLAGS bits 8.. switch pieces of the row-block kernel off (tools/ln_bench.py)
#define LN_DBG(pd, b) ((pd).flags & (b))
#else
#define LN_DBG(pd, b) false
#endif
template <int MODE, bool PEND>
__global__ __launch_bounds__(512) void ln_row_block_kernel(float* __restrict__ x, int ldx, f16* __restrict__ out, int M, int D,
                                                           const float* __restrict__ p0, const float* __restrict__ p1,
                                                           int mod_stride, const int* __restrict__ rows, int rows_per_mod,
                                                           LnPending pd) {
    __shared__ float red[16];
    // A block lives ~3 us and most of it is latency, so the head is laid out by hand (round 3): the loads that come from memory / the Infinity Cache —
    // the residual row and the split-K slabs — are requested FIRST (they need five kernel arguments); the block-uniform table indices and the statistics'
    // shift follow as SCALAR loads (their own counter: in flight beside the vector loads); the per-frame vectors (L2 hits) are requested last, as soon as
    // the indices are back.  Before, hipcc had strung seven dependent s_load / s_waitcnt round trips in front of the first slab load.
    const int m = blockIdx.x;
    const int c = threadIdx.x * 4;
    const int nw = (blockDim.x + 63) >> 6, wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* xr = x + (size_t)m * ldx;
    const bool act = c < D;
    const bool nt = PEND && (pd.flags & 4);   // the row and the slabs are read ONCE: non-temporal loads keep them from pushing the next GEMM's prefetched
                                              // weight slice (docs/LABNOTES.md 4.10) out of the 4 MiB L2
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f}, av = v, bv = v;
    if (act) v = nt ? __builtin_nontemporal_load((const f32x4*)(xr + c)) : *(const f32x4*)(xr + c);
    f32x4 sl[8];
    int nchunk = 0;
    if (PEND && act) {
        // every load of the pending update is issued before the first add: the split-K slabs are independent, and a load-add-load-add chain costs
        // one memory round trip per slab.  Unused slots of a chunk re-read slab 0 (an L1 hit) and are not added; the order of the adds (bias, slab 0,
        // slab 1, ...) is fixed.
        const float* pp = pd.parts + (size_t)m * pd.ld + c;
        auto slab = [&](int sp) -> f32x4 {
            const f32x4* q = (const f32x4*)(pp + (size_t)sp * pd.slab_stride);
            return nt ? __builtin_nontemporal_load(q) : *q;
        };
        nchunk = pd.nsplit <= 1 ? 1 : pd.nsplit <= 2 ? 2 : pd.nsplit <= 4 ? 4 : 8;     // block-uniform
        if (nchunk == 1) {
            sl[0] = slab(0);
        } else if (nchunk == 2) {
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) sl[sp] = slab(sp < pd.nsplit ? sp : 0);
        } else if (nchunk == 4) {
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) sl[sp] = slab(sp < pd.nsplit ? sp : 0);
        } else {
#pragma unroll
            for (int sp = 0; sp < 8; ++sp) sl[sp] = slab(sp < pd.nsplit ? sp : 0);
        }
    }
    int row = 0, gr = 0;
    if (MODE == 0) {
        row = m / rows_per_mod;
        if (rows && !LN_DBG(pd, 0x2000)) row = rows[row];
    }
    if (PEND && pd.gate) {
        gr = m / pd.rows_per_gate;
        if (pd.gate_rows && !LN_DBG(pd, 0x2000)) gr = pd.gate_rows[gr];
    }
    const float kshift = LN_DBG(pd, 0x200) ? 0.f : xr[0];   // the shift of the one-pass statistics below (same cache line as thread 0's own load)
    if (act && !LN_DBG(pd, 0x100)) {
        av = *(const f32x4*)((MODE == 0 ? p1 + (size_t)row * mod_stride : p0) + c);
        bv = *(const f32x4*)((MODE == 0 ? p0 + (size_t)row * mod_stride : p1) + c);
    }
    if (PEND && act) {
        f32x4 y = pd.bias && !LN_DBG(pd, 0x100) ? *(const f32x4*)(pd.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 g4 = f32x4{1.f, 1.f, 1.f, 1.f};
        if (pd.gate && !LN_DBG(pd, 0x100)) g4 = *(const f32x4*)(pd.gate + (size_t)gr * pd.gate_stride + c);
#pragma unroll
        for (int sp = 0; sp < 8; ++sp)
            if (sp < nchunk && sp < pd.nsplit) y = y + sl[sp];
        if (pd.y_save) {   // training forward: the branch output before the gate (d gate needs it)
            float am = 0.f;
            *(f16x4*)(pd.y_save + (size_t)m * pd.ld + c) = sat4(y[0], y[1], y[2], y[3], am);
        }
        if (pd.gate) y = y * g4;
        v = v + y;
        // (the updated row is stored BEHIND the block barrier of the statistics below: in place it overwrites xr[0], which every wave of the
        // block reads as the shift K of the one-pass statistics — a wave that started late must not see the updated value there)
    }
    // Row statistics in ONE block-wide reduction: sums of (v - K) and (v - K)^2 with the shift K = the row's first residual element (a sample
    // of the row, so |mean - K| is of the order of the standard deviation and E[(v-K)^2] - E[v-K]^2 does not cancel), reduced side by side.
    // The two-pass form (mean, then centred squares) cost a second wave reduction + LDS exchange + barrier on the critical path of a
    // latency-bound launch (65 per forward).
    const float a0 = act ? v[0] - kshift : 0.f, a1 = act ? v[1] - kshift : 0.f, a2 = act ? v[2] - kshift : 0.f, a3 = act ? v[3] - kshift : 0.f;
    const float s1 = wave_sum_dpp((a0 + a1) + (a2 + a3));
    const float s2 = wave_sum_dpp((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3));
    if (lane == 0) { red[wid] = s1; red[8 + wid] = s2; }
    if (!LN_DBG(pd, 0x400)) __syncthreads();   // every wave has consumed its copy of kshift = xr[0] by now
    if (PEND && act) {
        float* xw = pd.x_out ? pd.x_out + (size_t)m * ldx + c : xr + c;   // training forward keeps every residual state
        if (LN_DBG(pd, 0x800)) {}
        else if (pd.flags & 1) store16_sc1(xw, v);
        else *(f32x4*)xw = v;
    }
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { t1 += i < nw ? red[i] : 0.f; t2 += i < nw ? red[8 + i] : 0.f; }   // red[] is read with ds_read_b128s, not a counted loop
    const float m1 = t1 / (float)D, mean = kshift + m1;
    const float var = fmaxf(t2 / (float)D - m1 * m1, 0.f);
    const float rstd = 1.0f / sqrtf(var + 1e-6f);
    if (!act) return;
    const float dd[4] = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
    float yv[4], amax = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xh = dd[e] * rstd;
        if (MODE == 0) {
            const float sc = av[e] + 1e-6f;
            yv[e] = xh * (1.0f + sc) + bv[e];
        } else {
            yv[e] = xh * av[e] + bv[e];
        }
    }
    int mo = m;
    if (pd.tperm_T > 0) {   // block-uniform: (b, t, p) -> (b, p / 16, t, p % 16), see LnPending
        const int fr = m / pd.tperm_P, pp = m - fr * pd.tperm_P, bb = fr / pd.tperm_T, tt = fr - bb * pd.tperm_T;
        mo = ((bb * (pd.tperm_P >> 4) + (pp >> 4)) * pd.tperm_T + tt) * 16 + (pp & 15);
    }
    if (!LN_DBG(pd, 0x1000)) store_f16x4_paired<1>(out + tiled_off(mo, c, D), sat4(yv[0], yv[1], yv[2], yv[3], amax), lane, pd.flags & 2);
    else if (yv[0] == 1.2345f) out[0] = (f16)yv[1];
    sat_report(amax, pd.err_flag);
}

// ------------------------------------------------------------------------------------------
__global__ void patchify_kernel(const float* __restrict__ img, const int* __restrict__ frame_index, int NB, int C, int H,
                                int W, int p, f16* __restrict__ out, int ldo, float a, float b, int* err_flag) {
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
            // the model's inputs are where a NaN / inf can enter the fp16 pipeline (inside it every store saturates)
            if (!(__builtin_fabsf(val) <= F16_MAX)) {
                if (err_flag) atomicOr(err_flag, val != val || __builtin_isinf(val) ? ERR_NONFINITE : ERR_F16_SAT);
                if (val == val) val = __builtin_amdgcn_fmed3f(val, -F16_MAX, F16_MAX);
            }
        }
        out[tiled_off((int)m, k, ldo)] = (f16)val;
    }
}

// The same for patch sizes that are multiples of 4 (the VAE's 20 x 20, the DiT's 2 x 2 stays on the kernel above): one block per token, a thread per FOUR
// consecutive pixels of a patch row = one 16-byte load and one 8-byte store of the tile-major operand (k % 4 == 0 keeps them inside one 16-byte chunk);
// the token's image coordinates are computed once per block.  The scalar kernel moved 333 MB in 253 us for the trainer's 80 frames (round 5).
__global__ __launch_bounds__(256) void patchify4_kernel(const float* __restrict__ img, const int* __restrict__ frame_index, int C, int H, int W, int p,
                                                      f16* __restrict__ out, int ldo, float a, float b, int* err_flag) {
    const int gh = H / p, gw = W / p, Kp = C * p * p, pp = p * p;
    const int m = blockIdx.x;
    const int x = m % gw, y = (m / gw) % gh, nb = m / (gw * gh);
    const int f = frame_index ? frame_index[nb] : nb;
    const float* base = img + ((size_t)f * C * H + (size_t)y * p) * W + x * p;
    int flags = 0;
    for (int k = 4 * threadIdx.x; k < ldo; k += 4 * blockDim.x) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (k < Kp) {
            const int c = k / pp, rem = k - c * pp, ph = rem / p, pw = rem - ph * p;
            const float4 t = *(const float4*)(base + ((size_t)c * H + ph) * W + pw);
            v[0] = a * t.x + b; v[1] = a * t.y + b; v[2] = a * t.z + b; v[3] = a * t.w + b;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (!(__builtin_fabsf(v[e]) <= F16_MAX)) {   // the model's inputs are where a NaN / inf can enter the fp16 pipeline
                    flags |= (v[e] != v[e] || __builtin_isinf(v[e])) ? ERR_NONFINITE : ERR_F16_SAT;
                    if (v[e] == v[e]) v[e] = __builtin_amdgcn_fmed3f(v[e], -F16_MAX, F16_MAX);
                }
        }
        *(f16x4*)(out + tiled_off(m, k, ldo)) = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
    }
    if (flags && err_flag) atomicOr(err_flag, flags);
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
                                       int Cp, float scale, int tiled, int* err_flag) {
    const size_t total = (size_t)Rp * Cp;
    bool sat = false;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % Cp);
        const size_t r = idx / Cp;
        float v = 0.f;
        if (r < (size_t)R && c < C) v = src[r * lds + c] * scale;
        if (v == v) {   // weights / latents beyond the operand type's range saturate — and say so (a weight clamped in silence would be a wrong model)
            sat |= __builtin_fabsf(v) > F16_MAX;
            v = __builtin_amdgcn_fmed3f(v, -F16_MAX, F16_MAX);
        }
        dst[tiled ? tiled_off((int)r, c, Cp) : idx] = (f16)v;
    }
    if (sat && err_flag) atomicOr(err_flag, ERR_F16_SAT);
}

__global__ void copy_f32_kernel(const float* __restrict__ src, int lds, int R, int C, float* __restrict__ dst, int ldd, int c0) {
    const size_t total = (size_t)R * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const size_t r = idx / C;
        dst[r * ldd + c0 + c] = src[r * lds + c];
    }
}

__global__ void rope_interleave_kernel(const float* cos_t, const float* sin_t, float* cs, int npos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npos * 32) {
        const int pos = i >> 5, k = i & 31;
        cs[(size_t)pos * 64 + 2 * k] = cos_t[(size_t)pos * 64 + 2 * k];
        cs[(size_t)pos * 64 + 2 * k + 1] = sin_t[(size_t)pos * 64 + 2 * k];
    }
}

__global__ void fill_f32_kernel(float* dst, size_t n, float v) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) dst[idx] = v;
}

__global__ void axpy_f32_kernel(float* y, const float* x, float alpha, size_t n) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) y[idx] += alpha * x[idx];
}

__global__ void add_f32_kernel(const float* a, const float* b, float* out, size_t n) {
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x)
        out[idx] = a[idx] + b[idx];
}

__global__ void step_setup_kernel(StepParams* dst, StepParams v, int* frame_idx, int* mod_rows, int* last_rows, int* changed, int B, int Tq, int T,
                                  int F, int use_cur) {
    if (threadIdx.x == 0) *dst = v;
    const int first = use_cur ? v.cur : v.first;
    for (int i = threadIdx.x; i < B * Tq; i += blockDim.x) {
        const int b = i / Tq, tl = i - b * Tq;
        frame_idx[i] = b * F + first + tl;
        if (v.cond_step >= 0) {
            const int tw = use_cur ? T - 1 : tl;   // position inside the window
            const int r = tw < T - 1 ? b * (T - 1) + tw : B * (T - 1) + v.cond_step * B + b;
            mod_rows[i] = r;
            if (last_rows) {      // which slots of the current-step table (gather_rows_kernel) hold another row than this step needs
                changed[i] = last_rows[i] != r;
                last_rows[i] = r;
            }
        }
    }
}

// Current-step conditioning table: slot i <- row rows[i] of the per-frame table, for the slots step_setup flagged.  The LayerNorms and gates of
// the step then index the modulation by frame slot directly; through the row indirection every one of the 65 LayerNorm launches of a forward
// paid a dependent load (index, then the row's shift / scale) at the head of its critical path: 5.53 us per launch in the sampler against
// 4.87 us in the plain forward.  Per step only the denoised frame's row changes: one 0.8 MB copy.
// (W2 > 0: a second table — the c1 / c2 tables of the LayerNorm fold — is gathered by the same launch: blocks beyond W cover it)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, const int* __restrict__ rows, const int* __restrict__ changed,
                                                          float* __restrict__ cur, int W, const float* __restrict__ table2, float* __restrict__ cur2, int W2) {
    const int slot = blockIdx.y;
    if (!changed[slot]) return;
    const int nb1 = (W + 1023) >> 10;
    if ((int)blockIdx.x < nb1) {
        const size_t c = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
        if (c >= (size_t)W) return;
        *(f32x4*)(cur + (size_t)slot * W + c) = *(const f32x4*)(table + (size_t)rows[slot] * W + c);
    } else {
        const size_t c = ((size_t)(blockIdx.x - nb1) * 256 + threadIdx.x) * 4;
        if (c >= (size_t)W2) return;
        *(f32x4*)(cur2 + (size_t)slot * W2 + c) = *(const f32x4*)(table2 + (size_t)rows[slot] * W2 + c);
    }
}

// X operands of the grouped table GEMM of the LayerNorm fold (gemm.h launch_gemm_grouped): group g <- columns [col[g], col[g] + D) of the fp32
// modulation table `mod` [R][MODW] as fp16 TILE-MAJOR [Rp][D] (rows >= R zero); scale groups (is_scale[g]) hold 1 + (scale + 1e-6), the factor
// the producer epilogue applies to the residual (model/dit.py:19-27 modulate).
__global__ __launch_bounds__(256) void ctab_inputs_kernel(const float* __restrict__ mod, int MODW, int R, int Rp, int D, const int* __restrict__ col,
                                                          const int* __restrict__ is_scale, f16* __restrict__ sx, size_t group_stride) {
    const int g = blockIdx.y;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;      // one thread per 8 consecutive k of one row
    const int per_row = D >> 3;
    const int r = (int)(idx / per_row), k = (int)(idx - (size_t)r * per_row) * 8;
    if (r >= Rp) return;
    union { f16x8 h; u32x4 u; } o;
    if (r < R) {
        const float* src = mod + (size_t)r * MODW + col[g] + k;
        const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
        const bool sc = is_scale[g] != 0;
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = sc ? 1.0f + (v[e] + 1e-6f) : v[e];
            x = __builtin_amdgcn_fmed3f(x, -F16_MAX, F16_MAX);
            o.h[e] = (f16)x;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o.h[e] = (f16)0.f;
    }
    *(u32x4*)(sx + (size_t)g * group_stride + tiled_off(r, k, D)) = o.u;
}

__global__ void cond_inputs_frame_kernel(int rows, int B, int T, int F, int start, int cur, int t_ctx, const int* __restrict__ t_steps,
                                         const float* __restrict__ sincos, float* __restrict__ E,
                                         const float* __restrict__ actions, int A, float* __restrict__ HC, int ldhc, int D,
                                         int Apad, int* err_flag) {
    const int r = blockIdx.x;
    if (r >= rows) return;
    const int nctx = B * (T - 1);
    int t, b, frame;
    if (r < nctx) {
        b = r / (T - 1);
        frame = start + (r - b * (T - 1));
        t = t_ctx;
    } else {
        const int q = r - nctx;
        const int sidx = q / B;
        b = q - sidx * B;
        frame = cur;
        t = t_steps[sidx];
    }
    if (t < 0 || t > 999) {
        if (threadIdx.x == 0 && err_flag) atomicOr(err_flag, 1);
        t = t < 0 ? 0 : 999;
    }
    for (int j = threadIdx.x; j < 256; j += blockDim.x) E[(size_t)r * 256 + j] = sincos[(size_t)t * 256 + j];
    const float* arow = actions ? actions + ((size_t)b * F + frame) * A : nullptr;
    for (int j = threadIdx.x; j < Apad; j += blockDim.x) HC[(size_t)r * ldhc + D + j] = (arow && j < A) ? arow[j] : 0.f;
}

__global__ void cond_inputs_kernel(const int64_t* __restrict__ t64, int rows, int Tq, const StepParams* __restrict__ sp, int use_cur,
                                   const float* __restrict__ sincos, float* __restrict__ E, const float* __restrict__ actions,
                                   long long act_outer, long long act_inner, int A, float* __restrict__ HC, int ldhc, int D,
                                   int Apad, int* err_flag) {
    const int r = blockIdx.x;
    if (r >= rows) return;
    const int ro = r / Tq, ri = r - ro * Tq;
    long long t;
    int frame0 = 0;
    if (t64) {
        t = (long long)t64[r];
    } else {
        t = ri == Tq - 1 ? sp->t_cur : sp->t_ctx;
        frame0 = use_cur ? sp->cur : sp->first;
    }
    if (t < 0 || t > 999) {
        if (threadIdx.x == 0 && err_flag) atomicOr(err_flag, 1);
        t = t < 0 ? 0 : 999;
    }
    for (int j = threadIdx.x; j < 256; j += blockDim.x) E[(size_t)r * 256 + j] = sincos[(size_t)t * 256 + j];
    const float* arow = actions ? actions + ro * act_outer + (frame0 + ri) * act_inner : nullptr;
    for (int j = threadIdx.x; j < Apad; j += blockDim.x) HC[(size_t)r * ldhc + D + j] = (arow && j < A) ? arow[j] : 0.f;
}

__device__ __forceinline__ float ddim_one(float xc, float vp, float at, float an, int is_final) {
    const float x0 = sqrtf(at) * xc - sqrtf(1.0f - at) * vp;
    if (is_final) return x0;
    const float eps = (sqrtf(1.0f / at) * xc - x0) / sqrtf(1.0f / at - 1.0f);
    return sqrtf(an) * x0 + sqrtf(1.0f - an) * eps;
}

__global__ void ddim_update_kernel(const float* __restrict__ x, size_t x_stride, const float* __restrict__ v, size_t v_stride,
                                   float* __restrict__ out, size_t out_stride, int n, const float* __restrict__ alpha_t,
                                   const float* __restrict__ alpha_next, int is_final) {
    const int b = blockIdx.y;
    const float at = alpha_t[b];
    const float an = alpha_next ? alpha_next[b] : 1.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[b * out_stride + i] = ddim_one(x[b * x_stride + i], v[b * v_stride + i], at, an, is_final);
}

__global__ void ddim_update_step_kernel(float* __restrict__ x, size_t frames_per_sample, const float* __restrict__ v,
                                        size_t v_stride, int n, const StepParams* __restrict__ sp) {
    const int b = blockIdx.y;
    const float at = sp->alpha_t, an = sp->alpha_next;
    const int fin = sp->is_final;
    float* xf = x + ((size_t)b * frames_per_sample + sp->cur) * n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        xf[i] = ddim_one(xf[i], v[b * v_stride + i], at, an, fin);
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

__global__ void unpad_f16_kernel(const f16* __restrict__ src, int lds, int R, int C, float* __restrict__ dst, int tiled) {
    const size_t total = (size_t)R * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const size_t r = idx / C;
        dst[idx] = (float)src[tiled ? tiled_off((int)r, c, lds) : r * lds + c];
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

// ------------------------------------------------------------------------------------------
// Frame ingest (web_dataset.py:41-57,105-107 / hf_dataset.py:22-41 / generate.py:150-153): ToTensor (u8 / 255, HWC -> CHW),
// SplitImages (n frames side by side along W) and Resize((OH, OW)) — bilinear with antialiasing as torch's
// F.interpolate(mode="bilinear", antialias=True, align_corners=False), which torchvision's tensor Resize calls: per axis
//   scale = in / out, support = max(scale, 1), centre = scale (i + 0.5), taps j in [int(centre - support + 0.5), int(centre + support
//   + 0.5)) clipped to the image, weight = max(0, 1 - |j + 0.5 - centre| / max(scale, 1)), normalised to sum 1.
// Up-scaling (the dataset's 270 x 480 -> 360 x 640) degenerates to plain bilinear with at most 2 x 2 taps.  One thread per output
// pixel evaluates the separable filter directly (sum_y wy sum_x wx); HBM-bound, coalesced along ox.
// U8SRC: source is a uint8 HWC strip (H, n W, 3) holding n frames; else float NCHW (n, 3, H, W).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void aa_taps(int i, float scale, int in_size, int& lo, int& cnt, float& centre, float& inv) {
    const float support = scale >= 1.0f ? scale : 1.0f;
    centre = scale * ((float)i + 0.5f);
    inv = scale >= 1.0f ? 1.0f / scale : 1.0f;
    lo = (int)(centre - support + 0.5f);
    lo = lo < 0 ? 0 : lo;
    int hi = (int)(centre + support + 0.5f);
    hi = hi > in_size ? in_size : hi;
    cnt = hi - lo;
}
__device__ __forceinline__ float aa_w(int j, float centre, float inv) {
    float x = ((float)j - centre + 0.5f) * inv;
    x = x < 0.f ? -x : x;
    return x < 1.0f ? 1.0f - x : 0.0f;
}
template <bool U8SRC>
__global__ void resize_aa_kernel(const void* __restrict__ src, float* __restrict__ dst, int n, int H, int W, int OH, int OW, float in_scale) {
    const size_t total = (size_t)n * 3 * OH * OW;
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % OW), oy = (int)((idx / OW) % OH), c = (int)((idx / ((size_t)OW * OH)) % 3);
        const int f = (int)(idx / ((size_t)OW * OH * 3));
        int y0, ny, x0, nx;
        float cy, iy, cx, ix;
        aa_taps(oy, sy, H, y0, ny, cy, iy);
        aa_taps(ox, sx, W, x0, nx, cx, ix);
        float wxs = 0.f;
        for (int j = 0; j < nx; ++j) wxs += aa_w(x0 + j, cx, ix);
        float acc = 0.f, wys = 0.f;
        for (int a = 0; a < ny; ++a) {
            const float wy = aa_w(y0 + a, cy, iy);
            wys += wy;
            float row = 0.f;
            for (int j = 0; j < nx; ++j) {
                float v;
                if (U8SRC) v = (float)((const uint8_t*)src)[((size_t)(y0 + a) * ((size_t)n * W) + (size_t)f * W + (x0 + j)) * 3 + c];
                else v = ((const float*)src)[(((size_t)f * 3 + c) * H + (y0 + a)) * W + (x0 + j)];
                row += aa_w(x0 + j, cx, ix) * v;
            }
            acc += wy * row;
        }
        dst[idx] = acc / (wxs * wys) * in_scale;
    }
}

inline int grid_for(size_t total, int block = 256) {
    size_t g = (total + block - 1) / block;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

// to_qkv weight rows [q D | k D | v D] (tile-major fp16, K = D) -> head-major rows: 16-byte chunks keep their place inside the row (both row indices are
// congruent mod 8, so the tile swizzle maps chunk to chunk).  mode 0: [head][q 64 | k 64 | v 64] (the fused temporal kernel's W); mode 1: [head][wave w of 4][q | k | v
// features 16 w .. 16 w + 15] (the fused spatial kernel's W: every compute wave holds the same 16 head features of q, k and v)
__global__ void qkv_head_major_kernel(const f16* __restrict__ src, f16* __restrict__ dst, int D, int mode) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int cpr = D >> 3;
    if (i >= (size_t)3 * D * cpr) return;
    const int nd = (int)(i / cpr), kc = (int)(i - (size_t)nd * cpr);
    const int hd = nd / 192, rem = nd - hd * 192;
    int which, d;
    if (mode == 0) {
        which = rem >> 6;
        d = rem & 63;
    } else {
        const int w = rem / 48, r48 = rem - w * 48;
        which = r48 >> 4;
        d = 16 * w + (r48 & 15);
    }
    const int ns = which * D + hd * 64 + d;
    *(uint4*)(dst + tiled_off(nd, kc * 8, D)) = *(const uint4*)(src + tiled_off(ns, kc * 8, D));
}

// Timer calibration (api.hip gtav_timer_calibrate): ONE wave that spins on s_memrealtime (100 MHz, chip-wide) for `ticks` ticks and writes how long it
// really ran (device clock, first instruction to last) into dev_ticks[0].  Launched through GTAV_LAUNCH like every profiled kernel, so the difference between
// the event pair's reading and the device's own figure is the profiler's constant offset (marker packet + completion-signal handling).
__global__ void calib_spin_kernel(unsigned long long ticks, unsigned long long* __restrict__ dev_ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t1 = t0;
    while (t1 - t0 < ticks) {
        __builtin_amdgcn_s_sleep(2);
        t1 = __builtin_amdgcn_s_memrealtime();
    }
    if (threadIdx.x == 0) dev_ticks[0] = t1 - t0;
}

}  // namespace

// -DGTAV_EXPERIMENTS builds: GTAV_LN_FLAGS (see LnPending::flags)
static int g_ln_flags = GTAV_ENV_INT("GTAV_LN_FLAGS", 7);   // default: all stores written through (B = 1: LN 0.57 -> 0.54 ms per forward)

// ln_wave_row_kernel from this many rows on when there is no pending update (every such launch: 3.4 vs 3.6 us at 720 rows, 7.1 vs 8.9 at 5 760, 45 vs 67 at 46 080)
static int g_ln_wave_row_min = GTAV_ENV_INT("GTAV_LN_WAVE_ROW_MIN", 0);
// (Round 2 had tried one wave per row for the batch-1 forward's launches WITH a pending update: 5.2-5.3 us per launch against 4.9 for the row-block
// kernel at M = 720 in a one-process A/B, profiles/round2/forward_ab_B1_ln_wave_row.txt; round 5 measured the same again, see above.)

#define LN_WAVE_ROW_(MODE, NV, P0, P1, STRIDE, ROWS, RPM) \
    GTAV_LAUNCH((ln_wave_row_kernel<MODE, NV>), dim3(cdiv(M, 4)), dim3(256), 0, stream, x, ldx, out, M, D, P0, P1, STRIDE, ROWS, RPM, pd_.flags, err_flag)   /* (GTAV_LAUNCH: the profiler's events ride on the dispatch) */
#define LN_DISPATCH(MODE, P0, P1, STRIDE, ROWS, RPM)                                                     \
    do {                                                                                                  \
        LnPending pd_;                                                                                    \
        memset(&pd_, 0, sizeof(pd_));                                                                     \
        if (pend) pd_ = *pend;                                                                            \
        pd_.flags = g_ln_flags;                                                                           \
        pd_.err_flag = err_flag;                                                                          \
        GTAV_REQUIRE(pd_.tperm_T == 0 || (pd_.tperm_P % 16 == 0 && M % (pd_.tperm_T * pd_.tperm_P) == 0), "ln: bad output row permutation"); \
        const bool upd_ = pend && pend->parts;   /* a LnPending without slabs carries the output row permutation only */ \
        if (!upd_ && pd_.tperm_T == 0 && M >= g_ln_wave_row_min && (D == 256 || D == 512 || D == 1024 || D == 2048)) { \
            if (D == 256) LN_WAVE_ROW_(MODE, 1, P0, P1, STRIDE, ROWS, RPM);                               \
            else if (D == 512) LN_WAVE_ROW_(MODE, 2, P0, P1, STRIDE, ROWS, RPM);                          \
            else if (D == 1024) LN_WAVE_ROW_(MODE, 4, P0, P1, STRIDE, ROWS, RPM);                         \
            else LN_WAVE_ROW_(MODE, 8, P0, P1, STRIDE, ROWS, RPM);                                        \
        } else { /* one block per row */                                                                  \
            const dim3 g_(M), b_(round_up(D / 4, 64));                                                    \
            if (upd_) GTAV_LAUNCH((ln_row_block_kernel<MODE, true>), g_, b_, 0, stream, x, ldx, out, M, D, P0, P1, STRIDE, ROWS, RPM, pd_); \
            else GTAV_LAUNCH((ln_row_block_kernel<MODE, false>), g_, b_, 0, stream, x, ldx, out, M, D, P0, P1, STRIDE, ROWS, RPM, pd_);     \
        }                                                                                                 \
    } while (0)

int launch_ln_modulate(float* x, int ldx, f16* out, int ldo, int M, int D, const float* shift, const float* scale,
                       int mod_stride, const int* rows, int rows_per_mod, const LnPending* pend, int* err_flag, hipStream_t stream) {
    GTAV_REQUIRE(D % 64 == 0 && D <= 2048 && rows_per_mod > 0 && ldo == D, "ln_modulate: D=%d must be %%64, <= 2048, ldo == D", D);
    GTAV_REQUIRE(!pend || !pend->parts || (pend->nsplit >= 1 && pend->nsplit <= 8 && pend->ld % 4 == 0 && (!pend->gate || pend->rows_per_gate > 0)),
                 "ln_modulate: bad pending update");
    LN_DISPATCH(0, shift, scale, mod_stride, rows, rows_per_mod);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_ln_affine(float* x, int ldx, f16* out, int ldo, int M, int D, const float* gamma, const float* beta,
                     const LnPending* pend, int* err_flag, hipStream_t stream) {
    GTAV_REQUIRE(D % 64 == 0 && D <= 2048 && ldo == D, "ln_affine: D=%d must be %%64, <= 2048, ldo == D", D);
    GTAV_REQUIRE(!pend || !pend->parts || (pend->nsplit >= 1 && pend->nsplit <= 8 && pend->ld % 4 == 0 && (!pend->gate || pend->rows_per_gate > 0)),
                 "ln_affine: bad pending update");
    LN_DISPATCH(1, gamma, beta, 0, (const int*)nullptr, 1);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
#undef LN_DISPATCH
#undef LN_WAVE_ROW_

int launch_patchify(const float* img, const int* frame_index, int NB, int C, int H, int W, int p, f16* out, int ldo,
                    float a, float b, int* err_flag, hipStream_t stream) {
    GTAV_REQUIRE(H % p == 0 && W % p == 0 && ldo >= C * p * p, "patchify: bad geometry");
    const size_t total = (size_t)NB * (H / p) * (W / p) * ldo;
    if (p % 4 == 0 && W % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)img & 15) == 0 && (size_t)NB * (H / p) * (W / p) < (1u << 31)) {
        hipLaunchKernelGGL(patchify4_kernel, dim3(NB * (H / p) * (W / p)), dim3(256), 0, stream, img, frame_index, C, H, W, p, out, ldo, a, b, err_flag);
        GTAV_CHECK_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total)), dim3(256), 0, stream, img, frame_index, NB, C, H, W, p, out,
                       ldo, a, b, err_flag);
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

int launch_convert_pad_f16(const float* src, int lds, int R, int C, f16* dst, int Rp, int Cp, float scale, int tiled,
                           hipStream_t stream, int* err_flag) {
    GTAV_REQUIRE(Rp >= R && Cp >= C, "convert_pad: padded shape smaller than source");
    GTAV_REQUIRE(!tiled || (Rp % 128 == 0 && Cp % 64 == 0), "convert_pad: tile-major needs Rp %% 128 == 0 and Cp %% 64 == 0");
    hipLaunchKernelGGL(convert_pad_f16_kernel, dim3(grid_for((size_t)Rp * Cp)), dim3(256), 0, stream, src, lds, R, C, dst,
                       Rp, Cp, scale, tiled, err_flag);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_qkv_head_major(const f16* src, f16* dst, int D, hipStream_t stream, int mode) {
    GTAV_REQUIRE(D % 128 == 0 && (mode == 0 || mode == 1), "qkv_head_major: D=%d must be a multiple of 128, mode=%d 0 or 1", D, mode);
    hipLaunchKernelGGL(qkv_head_major_kernel, dim3(grid_for((size_t)3 * D * (D / 8))), dim3(256), 0, stream, src, dst, D, mode);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_copy_f32(const float* src, int lds, int R, int C, float* dst, int ldd, int c0, hipStream_t stream) {
    hipLaunchKernelGGL(copy_f32_kernel, dim3(grid_for((size_t)R * C)), dim3(256), 0, stream, src, lds, R, C, dst, ldd, c0);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_rope_interleave(const float* cos_t, const float* sin_t, float* cs, int npos, hipStream_t stream) {
    hipLaunchKernelGGL(rope_interleave_kernel, dim3(cdiv(npos * 32, 256)), dim3(256), 0, stream, cos_t, sin_t, cs, npos);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_fill_f32(float* dst, size_t n, float v, hipStream_t stream) {
    hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(n)), dim3(256), 0, stream, dst, n, v);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_axpy_f32(float* y, const float* x, float alpha, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(axpy_f32_kernel, dim3(grid_for(n)), dim3(256), 0, stream, y, x, alpha, n);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_calib_spin(unsigned long long ticks, unsigned long long* dev_ticks, hipStream_t stream) {
    GTAV_LAUNCH(calib_spin_kernel, dim3(1), dim3(64), 0, stream, ticks, dev_ticks);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_add_f32(const float* a, const float* b, float* out, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(add_f32_kernel, dim3(grid_for(n)), dim3(256), 0, stream, a, b, out, n);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_step_setup(StepParams* dst, const StepParams& v, int* frame_idx, int* mod_rows, int* last_rows, int* changed, int B, int Tq, int T, int F,
                      int use_cur, hipStream_t stream) {
    hipLaunchKernelGGL(step_setup_kernel, dim3(1), dim3(64), 0, stream, dst, v, frame_idx, mod_rows, last_rows, changed, B, Tq, T, F, use_cur);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_gather_rows(const float* table, const int* rows, const int* changed, float* cur, int slots, int W, const float* table2, float* cur2, int W2,
                       hipStream_t stream) {
    GTAV_REQUIRE(W % 4 == 0 && W2 % 4 == 0 && slots > 0, "gather_rows: W=%d W2=%d slots=%d", W, W2, slots);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(W, 1024) + (table2 ? cdiv(W2, 1024) : 0), slots), dim3(256), 0, stream, table, rows, changed, cur, W,
                       table2, cur2, table2 ? W2 : 0);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_ctab_inputs(const float* mod, int MODW, int R, int Rp, int D, const int* col, const int* is_scale, int n_groups, f16* sx, size_t group_stride,
                       hipStream_t stream) {
    GTAV_REQUIRE(D % 64 == 0 && Rp % 128 == 0 && R <= Rp && n_groups > 0, "ctab_inputs: bad geometry");
    const size_t threads = (size_t)Rp * (D >> 3);
    hipLaunchKernelGGL(ctab_inputs_kernel, dim3((unsigned)((threads + 255) / 256), n_groups), dim3(256), 0, stream, mod, MODW, R, Rp, D, col, is_scale, sx,
                       group_stride);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_cond_inputs_frame(int rows, int B, int T, int F, int start, int cur, int t_ctx, const int* t_steps, const float* sincos,
                             float* E, const float* actions, int A, float* HC, int ldhc, int D, int Apad, int* err_flag,
                             hipStream_t stream) {
    hipLaunchKernelGGL(cond_inputs_frame_kernel, dim3(rows), dim3(256), 0, stream, rows, B, T, F, start, cur, t_ctx, t_steps, sincos,
                       E, actions, A, HC, ldhc, D, Apad, err_flag);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_cond_inputs(const int64_t* t64, int rows, int Tq, const StepParams* sp, int use_cur, const float* sincos, float* E,
                       const float* actions, int64_t act_outer, int64_t act_inner, int A, float* HC, int ldhc, int D, int Apad,
                       int* err_flag, hipStream_t stream) {
    GTAV_REQUIRE(rows > 0 && Tq > 0 && (t64 || sp), "cond_inputs: bad rows/Tq/timesteps");
    hipLaunchKernelGGL(cond_inputs_kernel, dim3(rows), dim3(256), 0, stream, t64, rows, Tq, sp, use_cur, sincos, E, actions,
                       (long long)act_outer, (long long)act_inner, A, HC, ldhc, D, Apad, err_flag);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_ddim_update(const float* x, size_t x_stride, const float* v, size_t v_stride, float* out, size_t out_stride,
                       int B, int n, const float* alpha_t, const float* alpha_next, int is_final, hipStream_t stream) {
    hipLaunchKernelGGL(ddim_update_kernel, dim3(cdiv(n, 256), B), dim3(256), 0, stream, x, x_stride, v, v_stride, out,
                       out_stride, n, alpha_t, alpha_next, is_final);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

int launch_ddim_update_step(float* x, size_t frames_per_sample, const float* v, size_t v_stride, int B, int n,
                            const StepParams* sp, hipStream_t stream) {
    hipLaunchKernelGGL(ddim_update_step_kernel, dim3(cdiv(n, 256), B), dim3(256), 0, stream, x, frames_per_sample, v, v_stride,
                       n, sp);
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

int launch_unpad_f16_to_f32(const f16* src, int lds, int R, int C, float* dst, int tiled, hipStream_t stream) {
    hipLaunchKernelGGL(unpad_f16_kernel, dim3(grid_for((size_t)R * C)), dim3(256), 0, stream, src, lds, R, C, dst, tiled);
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
int launch_resize_aa(const void* src, int src_is_u8_strip, float* dst, int n, int H, int W, int OH, int OW, hipStream_t stream) {
    GTAV_REQUIRE(src && dst && n >= 1 && H >= 1 && W >= 1 && OH >= 1 && OW >= 1, "resize: bad arguments");
    const size_t total = (size_t)n * 3 * OH * OW;
    if (src_is_u8_strip) hipLaunchKernelGGL(resize_aa_kernel<true>, dim3(grid_for(total)), dim3(256), 0, stream, src, dst, n, H, W, OH, OW, 1.0f / 255.0f);
    else hipLaunchKernelGGL(resize_aa_kernel<false>, dim3(grid_for(total)), dim3(256), 0, stream, src, dst, n, H, W, OH, OW, 1.0f);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}
int launch_latents_to_tokens(const float* lat, float* z, int N, int hw, int latent, hipStream_t stream) {
    hipLaunchKernelGGL(latents_to_tokens_kernel, dim3(grid_for((size_t)N * hw * latent)), dim3(256), 0, stream, lat, z, N, hw, latent);
    GTAV_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace gtav
