// Launchers of the non-GEMM kernels (all enqueue on `stream`, return 0 on success, never sync).
#pragma once
#include "common.h"
#include "gemm.h"

namespace gtav {

// ---- skinny.hip --------------------------------------------------------------------------
int launch_skinny_f32(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int M, int N,
                      int K, int act_silu, hipStream_t stream);
int skinny_init();  // raises the dynamic-LDS limit of the skinny kernels (call once per process)

// ---- elementwise.hip ---------------------------------------------------------------------
// Per-noise-step scalars of the fused sampler step, kept in device memory so that a captured hipGraph of the
// step can be replayed with new values (written by a one-thread kernel ahead of the graph launch).
struct StepParams {
    int first;      // first frame of the window inside the latent buffer
    int cur;        // frame being denoised
    int t_ctx;      // timestep of the context frames (stabilization level)
    int t_cur;      // timestep of frame `cur`
    int is_final;   // noise_idx == 0: return x_start (train_dit.py:119-120)
    float alpha_t, alpha_next;
    int cond_step;  // >= 0: row set of the per-frame conditioning table prepared by gtav_dit_prepare_frame; -1: inline
};
// One launch ahead of the (possibly graph-replayed) step: *dst = v, frame_idx[b*Tq+tl] = b*F + first + tl
// (first = use_cur ? v.cur : v.first) and, when v.cond_step >= 0, the conditioning-table row of every processed frame:
// context frame (b, tl < T-1) -> b*(T-1)+tl, frame `cur` of sample b -> B*(T-1) + cond_step*B + b.
int launch_step_setup(StepParams* dst, const StepParams& v, int* frame_idx, int* mod_rows, int* last_rows, int* changed, int B, int Tq, int T, int F,
                      int use_cur, hipStream_t stream);
// current-step conditioning table: cur[slot] <- table[rows[slot]] (W floats) for the slots step_setup flagged in `changed`
// (table2 / cur2 / W2: optional second table gathered by the same launch — the c1 / c2 tables of the LayerNorm fold)
int launch_gather_rows(const float* table, const int* rows, const int* changed, float* cur, int slots, int W, const float* table2, float* cur2, int W2,
                       hipStream_t stream);
// LayerNorm fold (gemm.h EPI_*_FOLD): fp16 tile-major X operands [n_groups][Rp][D] of the grouped table GEMM from the fp32 modulation table:
// group g = columns [col[g], col[g] + D) of mod [R][MODW]; is_scale[g] != 0 stores 1 + (scale + 1e-6).  col / is_scale are device arrays.
int launch_ctab_inputs(const float* mod, int MODW, int R, int Rp, int D, const int* col, const int* is_scale, int n_groups, f16* sx, size_t group_stride,
                       hipStream_t stream);
// Conditioning inputs for a whole generated frame (rows laid out as above, n_steps row sets for frame `cur`).
int launch_cond_inputs_frame(int rows, int B, int T, int F, int start, int cur, int t_ctx, const int* t_steps, const float* sincos,
                             float* E, const float* actions, int A, float* HC, int ldhc, int D, int Apad, int* err_flag,
                             hipStream_t stream);

// Deferred residual update executed by the LayerNorm that follows a residual GEMM (model/dit.py:207-223):
//   x[m] += gate[row(m)] * (sum_s parts[s][m] + bias)      (gate == nullptr -> 1, i.e. the VAE's plain residual)
// `parts` are the split-K slabs written by gemm EPI_PARTIAL: slab s at parts + s * slab_stride, rows of ld floats.
struct LnPending {
    const float* parts;
    int nsplit;
    size_t slab_stride;
    int ld;
    const float* bias;
    const float* gate;
    int gate_stride;
    const int* gate_rows;
    int rows_per_gate;
    int flags;   // set by the launcher: bit 0 = residual write-back as sc1 stores, bit 1 = fp16 output as paired 16-byte sc1 stores, bit 2 = row and slabs by non-temporal loads
    int* err_flag;   // device error word (common.h ERR_F16_SAT is raised when the fp16 output saturated); may be null
    // training forward (api_train.hip gtav_dit_train_forward): the backward pass needs every intermediate residual state and every
    // branch output, so the updated row goes to x_out (same leading dimension as x) instead of in place, and the branch output
    // y = sum_s parts[s] + bias (before the gate) is kept as fp16 rows of ld elements in y_save.  Both may be null.
    float* x_out;
    f16* y_save;
    // Row order of the fp16 output (row-block kernel only).  tperm_T > 0: token row m = (b * tperm_T + t) * tperm_P + p is written
    // to row ((b * (tperm_P / 16) + p / 16) * tperm_T + t) * 16 + p % 16 — 16 positions x all frames of the window contiguous, the
    // X-tile order of the fused temporal QKV + attention GEMM (gemm.hip gemm_qkvt_attn_kernel).  0 = identity.
    int tperm_T, tperm_P;
};

// LayerNorm outputs are GEMM A-operands: fp16 TILE-MAJOR with logical row length D (buffer rows padded to 128).
// LayerNorm(eps=1e-6, no affine) + adaLN modulate -> fp16  (model/dit.py:19-27,163-181)
//   out[m] = LN(x[m]) * (1 + (scale[row] + 1e-6)) + shift[row],  row = rows ? rows[m / rows_per_mod] : m / rows_per_mod
// (pend->tperm_T / tperm_P select a permuted OUTPUT row order, see LnPending)
int launch_ln_modulate(float* x, int ldx, f16* out, int ldo, int M, int D, const float* shift, const float* scale,
                       int mod_stride, const int* rows, int rows_per_mod, const LnPending* pend, int* err_flag, hipStream_t stream);
// LayerNorm(eps=1e-6) with affine weight/bias -> fp16   (model/vae.py:139,146,174)
int launch_ln_affine(float* x, int ldx, f16* out, int ldo, int M, int D, const float* gamma, const float* beta,
                     const LnPending* pend, int* err_flag, hipStream_t stream);

// Non-overlapping patch gather (im2col of a k = s = p conv):  img (NB, C, H, W) f32 -> A fp16 TILE-MAJOR, logical [M][ldo],
// token m = (nb, gh, gw), column k = (c, ph, pw); value = a * img + b.  Columns [C p p, ldo) are zeroed.
// `frame_index` (optional, length NB) picks frame f = frame_index[nb] out of the source buffer (frame stride =
// C*H*W floats), which is how the sampler reads its sliding window in place.
int launch_patchify(const float* img, const int* frame_index, int NB, int C, int H, int W, int p, f16* out, int ldo,
                    float a, float b, int* err_flag, hipStream_t stream);
// Inverse scatter of the projection output.  order 0: features (ph, pw, c) (DiT, model/dit.py:328-341);
// order 1: features (c, ph, pw) (VAE, model/vae.py:279-304).  out (NB, C, H, W) f32 = a * y + b.
int launch_unpatchify(const float* y, int ldy, float* img, int NB, int C, int H, int W, int p, int order, float a,
                      float b, hipStream_t stream);

// fp32 -> fp16 with zero padding: src [R][C] (ld = lds) -> dst [Rp][Cp]
// tiled != 0: dst is tile-major (common.h tiled_off) with Rp % 128 == 0, Cp % 64 == 0
// err_flag (optional device word): ERR_F16_SAT is raised when a finite value beyond the operand type's range was clamped
int launch_convert_pad_f16(const float* src, int lds, int R, int C, f16* dst, int Rp, int Cp, float scale, int tiled,
                           hipStream_t stream, int* err_flag = nullptr);
// inverse of the above without padding (state_dict round trip): dst[r][c] = (float)src[r][c]
int launch_unpad_f16_to_f32(const f16* src, int lds, int R, int C, float* dst, int tiled, hipStream_t stream);
int launch_copy_f32_strided(const float* src, int lds, int R, int C, float* dst, int ldd, hipStream_t stream);
int launch_copy_rows_f32(const float* src, size_t src_stride, float* dst, size_t dst_stride, int rows, size_t n, hipStream_t stream);
// buf[m][c] = clamp(buf[m][c], lo, hi) for c in [c0, c1)
int launch_clamp_cols(float* buf, int M, int ld, int c0, int c1, float lo, float hi, hipStream_t stream);
int launch_frames_to_u8(const float* img, uint8_t* out, int N, int H, int W, hipStream_t stream);
int launch_moments_to_latents(const float* mom, float* lat, int N, int hw, int latent, int mom_ch, float scale, hipStream_t stream);
int launch_latents_to_tokens(const float* lat, float* z, int N, int hw, int latent, hipStream_t stream);
// Frame ingest: antialiased bilinear resize to (OH, OW) of n frames, either from a uint8 HWC strip (H, n*W, 3) with /255 (ToTensor +
// SplitImages + Resize of the dataset step) or from float NCHW (n, 3, H, W).  dst (n, 3, OH, OW) f32.
int launch_resize_aa(const void* src, int src_is_u8_strip, float* dst, int n, int H, int W, int OH, int OW, hipStream_t stream);
// fp32 strided copy with padding (used to build concatenated fp32 weights): dst[r][c0 + c] = src[r][c]
int launch_copy_f32(const float* src, int lds, int R, int C, float* dst, int ldd, int c0, hipStream_t stream);
// tile-major to_qkv weight [3 D][D] -> head-major row order [head][q 64 | k 64 | v 64] (the fused temporal QKV + attention GEMM's W)
int launch_qkv_head_major(const f16* src, f16* dst, int D, hipStream_t stream, int mode = 0);   // mode 1: the fused spatial kernel's wave-interleaved order (elementwise.hip)
int launch_fill_f32(float* dst, size_t n, float v, hipStream_t stream);
// cs[pos][k] = (cos[pos][2k], sin[pos][2k]) for k < 32: the GEMM epilogue's interleaved RoPE table
int launch_rope_interleave(const float* cos_t, const float* sin_t, float* cs, int npos, hipStream_t stream);
int launch_add_f32(const float* a, const float* b, float* out, size_t n, hipStream_t stream);
// timer calibration: one wave spins `ticks` ticks of s_memrealtime (100 MHz) and stores the ticks it really ran in dev_ticks[0]; launched through GTAV_LAUNCH
int launch_calib_spin(unsigned long long ticks, unsigned long long* dev_ticks, hipStream_t stream);
int launch_axpy_f32(float* y, const float* x, float alpha, size_t n, hipStream_t stream);   // y += alpha x (x may alias y)

// Conditioning inputs (model/dit.py:96-118,359-364) for `rows` (b, frame) pairs, row r = (r / Tq, r % Tq):
//   E[r][0:256] = sincos_table[t_r],  t_r = t64 ? t64[r] : (r % Tq == Tq - 1 ? t_cur : t_ctx)   (train_dit.py:64-91)
//   HC[r][D : D+Apad] = actions[(r / Tq) * act_outer + (r % Tq) * act_inner + 0:A]  (zeros when actions == nullptr)
// With `sp` (sampler step): t_ctx / t_cur come from *sp and the action row of (b, tl) is
// actions[b * act_outer + ((use_cur ? sp->cur : sp->first) + tl) * act_inner].
int launch_cond_inputs(const int64_t* t64, int rows, int Tq, const StepParams* sp, int use_cur, const float* sincos /*[1000][256]*/,
                       float* E, const float* actions, int64_t act_outer, int64_t act_inner, int A, float* HC, int ldhc,
                       int D, int Apad, int* err_flag, hipStream_t stream);

// DDIM-style v-prediction update of the newest frame (train_dit.py:110-125), per sample b:
//   x0 = sqrt(a_t) x - sqrt(1 - a_t) v;  eps = (sqrt(1/a_t) x - x0) / sqrt(1/a_t - 1);
//   out = final ? x0 : sqrt(a_n) x0 + sqrt(1 - a_n) eps
// x, v, out: n elements per sample with given sample strides (floats).
int launch_ddim_update(const float* x, size_t x_stride, const float* v, size_t v_stride, float* out, size_t out_stride,
                       int B, int n, const float* alpha_t, const float* alpha_next, int is_final, hipStream_t stream);
// sampler form: frame sp->cur of x (B, F, n) is updated in place from v (row stride v_stride); alphas / is_final from *sp
int launch_ddim_update_step(float* x, size_t frames_per_sample, const float* v, size_t v_stride, int B, int n,
                            const StepParams* sp, hipStream_t stream);

// Training-side noising, v-target and squared-error partial sums (train_dit.py:621-650).
int launch_add_noise(const float* x, const float* noise, const float* alpha /*[rows]*/, float* out, int rows, int n,
                     float clamp_abs, hipStream_t stream);
int launch_vtarget(const float* x, const float* noise, const float* alpha /*[rows]*/, float* vt, int rows, int n,
                   float clamp_abs, hipStream_t stream);
int launch_mse(const float* a, size_t a_stride, const float* b, size_t b_stride, int rows, int n, float* out_scalar,
               hipStream_t stream);

// ---- train.hip (backward pass + optimizer) ---------------------------------------------
// src tile-major logical [R][C] (C % 64 == 0) -> dst tile-major logical [C][round_up(R, 64)], zero K padding
int launch_transpose_tiled_f16(const f16* src, int R, int C, f16* dst, hipStream_t stream);
// fp32 row-major [R][C] -> fp16 tile-major of the transpose, logical [C][round_up(R, 64)] inside [round_up(C, 128)][...]
int launch_convert_T_f16(const float* src, int lds, int R, int C, f16* dst, hipStream_t stream);
int launch_gelu_tiled(const f16* u, f16* h, size_t n, hipStream_t stream);
int launch_gelu_bwd_tiled(const f16* dh, const f16* u, f16* du, size_t n, int* err_flag, hipStream_t stream);
int launch_ln_mod_bwd(const float* dxn, const float* x, const float* scale, int mod_stride, int rows_per_mod, int M, int D, float* dres, int accumulate,
                      float* stats, hipStream_t stream);
int launch_frame_reduce_ln(const float* dxn, const float* x, const float* stats, int frames, int P, int D, float* dshift, float* dscale, int mod_stride,
                           hipStream_t stream);
// launch_gelu_bwd_tiled + the column sums of its output (db[n] += sum_m du[m][n]) in one pass
int launch_gelu_bwd_tiled_colsum(const f16* dh, const f16* u, f16* du, int M, int N, float* db, float* ws, int* err_flag, hipStream_t stream);
// db == nullptr in the two fused launchers: the per-split / per-frame partial sums stay in ws and the caller adds them later, several bias gradients per launch
int gelu_bwd_colsum_splits(int M);
int launch_colsum_reduce_multi(const float* const* ws, float* const* db, const int* splits, const int* N, int njobs, hipStream_t stream);
// launch_ln_mod_bwd + launch_frame_reduce_ln in one pass over dxn and x (M = frames x P rows)
bool ln_bwd_fused_ok(int D);
size_t ln_bwd_fused_workspace(int frames, int P, int D);
int launch_ln_mod_bwd_fused(const float* dxn, const float* x, const float* scale, int mod_stride, int frames, int P, int D, float* dres, int accumulate,
                            float* dshift, float* dscale, float* part, hipStream_t stream);
int launch_gate_bwd(const float* dres, const float* gate, int mod_stride, int rows_per_mod, int M, int D, f16* dy_tiled, int* err_flag, hipStream_t stream);
int launch_frame_reduce_gate(const float* dres, const f16* y, int frames, int P, int D, float* dgate, int mod_stride, hipStream_t stream);
// the two above + db[n] += sum_m dy[m][n] in one pass over dres (M = frames x P rows; ws: frames x D floats)
int launch_gate_bwd_fused(const float* dres, const f16* y, const float* gate, int mod_stride, int frames, int P, int D, f16* dy_tiled, float* dgate, float* db,
                          float* ws, int* err_flag, hipStream_t stream);
// column sums in a fixed order (no float atomics): ws = colsum_workspace(M, N) floats of scratch for the per-row-split partial sums
size_t colsum_workspace(int M, int N);
int launch_colsum_tiled_f16(const f16* dy, int M, int N, float* db, float* ws, hipStream_t stream);     // db[n] += sum_m dy[m][n]
int launch_colsum_f32(const float* a, int lda, int M, int N, float* db, float* ws, hipStream_t stream);   // db[n] += sum_m a[m][n]
int launch_to_tiled_f16(const float* a, int M, int D, f16* out, int* err_flag, hipStream_t stream);
int launch_mse_bwd_patch(const float* vpred, const float* vtarget, int B, int T, int C, int H, int W, int p, float scale, f16* dfo, int ldf, int* err_flag,
                         hipStream_t stream);
int launch_attn_spatial_bwd(const f16* Q, const f16* K, const f16* Vt, const f16* dO, int NB, int heads, int S, int D, const float* rope_cs, f16* dqkv,
                            int* err_flag, hipStream_t stream);
int launch_attn_temporal_bwd(const f16* q, const f16* kv, const f16* dO, int B, int P, int D, int T, int Tmax, const float* rope_cs, f16* dqkv, int* err_flag,
                             hipStream_t stream);
int launch_silu(const float* x, int ldx, float* y, int ldy, int R, int C, hipStream_t stream);
int launch_silu_bwd(const float* dy, int lddy, const float* x, int ldx, float* dx, int lddx, int R, int C, hipStream_t stream);
int launch_gemm_tn_f32(const float* dY, int lddy, const float* X, int ldx, int R, int N, int K, float* dW, int lddw, hipStream_t stream);   // dW += dY^T X
int launch_gemm_nn_f32(const float* dY, int lddy, const float* W, int ldw, int R, int N, int K, float* dX, int lddx, hipStream_t stream);   // dX = dY W
// dSc [R][D] = dmod [R][MODW] x W_ada [MODW][D].  `part` = workspace of ada_bwd_dx_workspace(MODW, D, R) floats (fp32 MFMA path: per-chunk partial
// sums reduced in a fixed order); nullptr selects the VALU kernel, which accumulates with atomics.  dSc is overwritten.
size_t ada_bwd_dx_workspace(int MODW, int D, int R);
int launch_ada_bwd_dx(const float* dmod, int MODW, const float* W, int D, int R, float* dSc, float* part, hipStream_t stream);                         // dSc += dmod W_ada
// multi-tensor AdamW: one descriptor per parameter, one work item per 64 x 64 weight tile / 4096-element run (train.hip)
struct AdamParam {
    float* p; int ldp, R, C;               // fp32 master (GEMM weights: contiguous [R][C]; fp32 parameters: in place, leading dimension ldp)
    const float* g; float *m, *v;          // gradient (scaled), AdamW moments, contiguous [R][C]
    f16* w16; int Cp16;                    // GEMM weights: tile-major fp16 W (logical row length Cp16), else null
    f16* wT; int RpT;                      // tile-major fp16 W^T (logical row length RpT = round_up(R, 64)), or null
};
struct AdamItem { int param; unsigned start; };   // GEMM weight: tile index (row-major over 64 x 64 tiles); fp32 parameter: first element
// ctl [8] floats: [0] sum of squares of the scaled gradients, [1] step coefficient (0 = step skipped), [2] skipped steps, [3] unscaled gradient
// norm, [4] applied steps (the Adam step count), [5] / [6] bias corrections of the step being applied (written by clip_coef on the device)
int launch_adamw_multi(const AdamParam* params, const AdamItem* items, int n_items, const float* ctl, float lr, float beta1, float beta2, float eps, float wd,
                       hipStream_t stream);
int sumsq_parts(size_t n);                                                         // per-block partial sums written by launch_sumsq
int launch_sumsq(const float* g, size_t n, float* part, hipStream_t stream);
// adds the partial sums in order (ctl[0]); err_flag (optional): ERR_F16_SAT / ERR_NONFINITE in the handle's error word count as overflow
// (step skipped) and are cleared
// Data-parallel training: the overflow decision of the optimizer step must be the SAME on every rank.  A rank whose fp16 gradient stores saturated
// (ERR_F16_SAT / ERR_NONFINITE in its error word) writes +inf into one element `g` of its gradient arena at the end of its backward pass, before the
// gradient all-reduce (SUM): every rank then sees a non-finite norm and skips the step (train.hip clip_coef_kernel).
int launch_overflow_publish(const int* err_flag, float* g, hipStream_t stream);
// clears `bits` of the device error word (stream-ordered)
int launch_err_clear(int* err_flag, int bits, hipStream_t stream);
int launch_clip_coef(float* ctl, const float* part, int nparts, float inv_scale, float max_norm, float beta1, float beta2, int* err_flag, hipStream_t stream);
int launch_adamw(float* p, int ldp, int R, int C, const float* g, float* m, float* v, const float* ctl, float lr, float beta1, float beta2, float eps,
                 float wd, hipStream_t stream);

// ---- attention.hip -----------------------------------------------------------------------
// Full (non-causal) attention over S tokens per (nb, head), head_dim 64 (model/attention.py:127-129, model/vae.py:101).
// Q,K [nb][heads][S][64], Vt [nb][heads][64][S] fp16 (layouts written by the QKV GEMM epilogue);
// O logical [nb*S][heads*64] fp16, TILE-MAJOR (A-operand of the out-projection GEMM).
// q_prescaled: Q already carries the softmax scale in the exponent's unit, q / 8 * log2 e (written that way by the to_qkv epilogue through
// GemmParams::rope_cs_q); only sequences that run the flash kernel take it — attn_spatial_wants_prescaled_q(S) says which
int launch_attn_spatial(const f16* Q, const f16* K, const f16* Vt, f16* O, int NB, int heads, int S, hipStream_t stream, bool q_prescaled = false);
bool attn_spatial_wants_prescaled_q(int S);
constexpr float kAttnQScale = 0.125f * 1.4426950408889634f;   // 1 / sqrt(64) * log2(e)
// Causal attention over the frames of a window per (b, p, head) (model/attention.py:62-64).
// q [B*Tq*P][D] row-major for frames t0 .. t0+Tq-1; kv cache [B][Tmax][P][2][D]; O logical like q but TILE-MAJOR.
int launch_attn_temporal(const f16* q, const f16* kv, f16* O, int B, int P, int D, int Tq, int t0, int Tmax,
                         hipStream_t stream);

}  // namespace gtav
