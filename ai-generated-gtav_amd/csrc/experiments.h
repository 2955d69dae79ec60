/* Entry points that exist ONLY in libgtav_amd_exp.so (csrc/build.sh exp: -DGTAV_EXPERIMENTS), the build the timing tools
 * under tools/ load.  They are not part of the product ABI (include/gtav_amd.h) because they change results.
 * That build also reads the GTAV_* environment variables listed in DESIGN.md 1. */
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* bit 0: skip the LDS fills after the prologue; bit 1: skip LDS reads + MFMA (results become WRONG: timing only);
 * bit 3: plain instead of write-through (sc1) output stores; bit 4: unstaged QKV epilogue (bits 3, 4 keep results). */
void gtav_op_gemm_set_debug(int32_t bits);
/* Per-block timeline of the following GEMM launches: 8 x uint64 per block (csrc/gemm.h gemm_set_stamps); NULL = off. */
void gtav_op_gemm_set_stamps(void* buf_dev, int32_t max_blocks);
/* ---- LayerNorm fold (round 3: correct, measured slower than the separate LayerNorm launch at every size; docs/LABNOTES.md) ---- */
struct gtav_dit;
/* LayerNorm fold (docs/LABNOTES.md 4.7).  The LayerNorm + adaLN modulate between a residual GEMM and the GEMM that consumes its output
 * (model/dit.py:19-27, 200-225: out-proj -> fc1 = seam A, fc2 -> next to_qkv / final projection = seam B) can run inside the two GEMM
 * epilogues instead of as a launch of its own: the producer updates the residual in place and emits x (1 + scale) plus per-row partial
 * sums, the consumer applies (acc - mean c1) rstd + c2 with per-frame tables c1 / c2 built next to the adaLN table.  Same arithmetic up to
 * fp32 summation order and one fp16 rounding of a differently scaled operand.  mode 0 = never, 1 = a seam folds at >= min_tokens tokens
 * per forward (the default mode; the default thresholds are "never": on MI355X the folded path measured slower than the separate LayerNorm
 * launch at every size tried, docs/LABNOTES.md 4.7), 2 = every seam at every size.  min_tokens_a / _b < 0 keep the current thresholds.
 * The first call that can fold anything allocates the tables (hipMalloc + hipMemset; fails on geometries whose frames are not a multiple of
 * 16 and >= 64 tokens); every call drops the captured graphs and the prepared frame of the handle.  Never folded on handles with training
 * enabled or gtav_dit_set_fused_temporal on. */
int gtav_dit_set_fold(gtav_dit* h, int32_t mode, int32_t min_tokens_a, int32_t min_tokens_b);

/* The two halves of a folded LayerNorm seam (docs/LABNOTES.md 4.7; model/dit.py:19-27, 200-225) as the model runs them.
 * producer: resid[m][n] += gate[f][n] (sum_k x[m][k] w[n][k] + bias[n]) in place (f = m / tokens_per_frame, vectors of frame f at
 *   gate / next_scale + f * mod_stride); a_out (fp16 tile-major [round_up(M,128)][N]) = resid (1 + next_scale[f][n] + 1e-6); stats_out [M][N/64][2] =
 *   (sum, sum of squares) of the updated row over each 64-feature slot.  N % 64 == 0, tokens_per_frame % 16 == 0 and >= 64, M % tokens_per_frame == 0.
 * consumer: y[m][n] = (sum_k a[m][k] w[n][k] - mean_m c1[f][n]) rstd_m + c2[f][n] with mean / rstd of row m from stats (K / 64 slots), then epi 0:
 *   out f32 row-major [M][ldo]; epi 2: GELU-tanh, fp16 tile-major with logical row length ldo.  c1 / c2: row f at + f * ldc. */
int gtav_op_gemm_fold_producer(const void* x_f16_dev, const void* w_f16_dev, const float* bias_dev, float* resid_dev, int32_t M, int32_t N, int32_t K,
                               const float* gate_dev, const float* next_scale_dev, int32_t mod_stride, int32_t tokens_per_frame, void* a_out_f16_dev,
                               float* stats_out_dev, void* stream);
int gtav_op_gemm_fold_consumer(const void* a_f16_dev, const void* w_f16_dev, int32_t M, int32_t N, int32_t K, int32_t epi, const float* stats_dev,
                               const float* c1_dev, const float* c2_dev, int32_t ldc, int32_t tokens_per_frame, void* out_dev, int32_t ldo, void* stream);
#ifdef __cplusplus
}
#endif
