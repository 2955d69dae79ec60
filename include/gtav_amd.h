/* gtav_amd — C ABI of the MI355X (gfx950) implementation of the AI-Generated-GTAV hot path.
 *
 * The reference (ikergarcia1996/AI-Generated-GTAV) has no FFI layer: its boundary for this path is the
 * Python class API of model/dit.py and model/vae.py plus train_dit.denoise_step.
 * This header is the C-ABI a binding of that API sits on; each entry point cites the reference
 * interface it replaces.  `ai-generated-gtav_amd/` holds the ctypes binding that mirrors the
 * reference classes on top of it; INTEGRATION.md shows the reference-side stub.
 *
 * Conventions: every function returns 0 on success and a non-zero code on failure (message via
 * gtav_last_error()); nothing throws across the ABI.  All pointers named *_dev are device (HBM)
 * pointers owned by the caller (torch storage); handles own their weights and workspace.  `stream`
 * is a hipStream_t passed as void* (NULL = default stream).  Handles are not thread-safe and belong to the
 * device that was current when they were created.
 *
 * What allocates / synchronises (everything else only enqueues kernels on `stream` and is hipGraph-capturable):
 *   gtav_*_create            hipMalloc + hipMemset of weights and workspace (synchronous)
 *   gtav_*_destroy           hipFree, destroys captured graphs and the private capture stream
 *   gtav_*_finalize          synchronises `stream` twice (host-side table construction and upload in between)
 *   gtav_dit_set_schedule    synchronous hipMemcpy of 1000 floats
 *   gtav_dit_prepare_frame   synchronises `stream`, then a synchronous hipMemcpy of the host timestep array
 *   gtav_dit_denoise_step    the first replayed call of a new (shape, buffers) key creates a private non-blocking stream
 *                            (once per handle), captures the step on it and instantiates a hipGraph (host work only, nothing
 *                            executes during capture); later calls are one hipGraphLaunch.  gtav_dit_set_graph(h, 0) selects
 *                            plain launches, which never allocate
 *   gtav_dit_set_fused_temporal / gtav_dit_set_fused_spatial   first enable: hipMalloc of the head-major weight copies, device synchronise
 *   gtav_dit_forward         with gtav_dit_profile enabled only: creates events and synchronises at the end of the forward
 *   gtav_dit_check / gtav_vae_check / gtav_dit_autorange   copy the device error words back and synchronise `stream`
 *   gtav_dit_train_enable    hipMalloc + hipMemset of masters, optimizer state, saved-activation and backward workspace, two small
 *                            synchronous hipMemcpy (the multi-tensor AdamW tables)
 *   gtav_dit_train_stats     copies four floats back and synchronises `stream`
 *   gtav_dit_get_opt_step / gtav_dit_set_opt_step   copy the optimizer's control words and synchronise `stream`
 *   gtav_comm_unique_id / gtav_comm_init / gtav_comm_destroy   dlopen of librccl.so on first use; RCCL's own bootstrap (blocking)
 * The library reads no environment variables (RCCL, once loaded, reads its own NCCL_* / RCCL_* variables).
 */
#ifndef GTAV_AMD_H
#define GTAV_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* gtav_last_error(void);
/* Library/ABI version; bumped when a signature changes. */
int gtav_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * DiT  — replaces model/dit.py:228-376 (class DiT) and its sub-modules
 *        (model/attention.py:13-136, model/rotary_embedding_torch.py, timm Mlp).
 * ---------------------------------------------------------------------------------------------- */
typedef struct gtav_dit_config {
    /* DiT.__init__ arguments, model/dit.py:233-244 */
    int32_t input_h, input_w, patch_size, in_channels, hidden_size, depth, num_heads;
    float mlp_ratio;
    int32_t external_cond_dim, max_frames;
    /* capacity of the handle's workspace */
    int32_t max_batch;      /* largest B of a forward / denoise call */
    int32_t max_cond_rows;  /* rows of the conditioning (adaLN) table; >= max_batch * max_frames */
} gtav_dit_config;

typedef struct gtav_dit gtav_dit;

int gtav_dit_create(const gtav_dit_config* cfg, gtav_dit** out);
void gtav_dit_destroy(gtav_dit* h);
/* One call per reference state-dict entry (names of SURVEY.md §8(b), e.g. "blocks.3.t_attn.to_qkv.weight");
 * src_dev is the fp32 tensor in torch layout.  Replaces safetensors.torch.load_model(model, path)
 * (generate.py:32).  Rotary `freqs` entries are passed under "spatial_rotary_emb.freqs" /
 * "temporal_rotary_emb.freqs" (any alias is accepted). */
int gtav_dit_set_weight(gtav_dit* h, const char* name, const float* src_dev, int64_t numel, void* stream);
/* Builds derived tables (RoPE cos/sin, fused conditioning weights). Call after all set_weight calls;
 * fails if a parameter is missing. */
int gtav_dit_finalize(gtav_dit* h, void* stream);
/* Copies the repacked value of one parameter back out as fp32 (for state_dict round trips / tests). */
int gtav_dit_get_weight(gtav_dit* h, const char* name, float* dst_dev, int64_t numel, void* stream);

/* DiT.forward(x, t, external_cond) — model/dit.py:343-376.
 * x (B,T,C,H,W) f32, t (B,T) int64, actions (B,T,external_cond_dim) f32 or NULL, out like x. */
int gtav_dit_forward(gtav_dit* h, const float* x_dev, const int64_t* t_dev, const float* actions_dev, float* out_dev,
                     int32_t B, int32_t T, void* stream);

/* Diffusion schedule used by the fused sampler step: alphas_cumprod (generate.py:195-198), 1000 floats (host). */
int gtav_dit_set_schedule(gtav_dit* h, const float* alphas_cumprod_host, int32_t n);

/* train_dit.denoise_step (train_dit.py:30-125) fused with generate.py:220 ("update only the last frame"):
 * latents x (B, F, C, H, W) f32; the window is frames [start, cur]; context frames use timestep t_ctx,
 * frame `cur` uses t_cur; frame `cur` of x is overwritten with x_pred (final step when is_final != 0).
 * mode 0: recompute the whole window (what the reference does on every step).
 * mode 1: context-cached step — only frame `cur` is pushed through the network, context K/V of every
 *         temporal layer come from the cache left by the last mode-0 call on the same window
 *         (exact: context activations do not depend on frame `cur`).  The handle records which
 *         (B, F, start, cur, x_dev) the caches describe; a mode-1 call for anything else, or after a
 *         gtav_dit_forward (which overwrites the caches), fails instead of attending to stale K/V.
 * actions (B, F, external_cond_dim) or NULL.  v_out (B, C, H, W) optional: v_pred of frame `cur`.
 * cond_step: -1 computes the conditioning c = t_emb(t) + action and its adaLN projections inside the step (as
 *   DiT.forward does, model/dit.py:359-366); >= 0 takes them from the table built by gtav_dit_prepare_frame. */
int gtav_dit_denoise_step(gtav_dit* h, float* x_dev, int32_t B, int32_t F, int32_t start, int32_t cur, int32_t t_ctx,
                          int32_t t_cur, int32_t t_next, int32_t is_final, const float* actions_dev, int32_t mode,
                          int32_t cond_step, float* v_out_dev, void* stream);
/* The conditioning does not depend on x: for one generated frame (window [start, cur]) the adaLN table of every noise
 * step is computed at once — context rows with t_ctx, n_steps row sets for frame `cur` with t_steps_host[s] — so the
 * 0.8 GB of fp32 conditioning weights are streamed once per frame instead of once per step.  Same arithmetic, same
 * results; needs max_cond_rows >= B * (cur - start + n_steps).  A gtav_dit_forward or a cond_step = -1 step overwrites
 * the table: later cond_step >= 0 calls fail until prepare_frame is called again. */
int gtav_dit_prepare_frame(gtav_dit* h, int32_t B, int32_t F, int32_t start, int32_t cur, int32_t t_ctx,
                           const int32_t* t_steps_host, int32_t n_steps, const float* actions_dev, void* stream);

/* The fused step replays a captured hipGraph per (shape, buffers) key by default; this call switches to plain
 * launches (results are identical). */
int gtav_dit_set_graph(gtav_dit* h, int32_t enable);

/* Optional (default OFF): full-window steps / forwards of up to 256 (80-token tile, head) pairs — batch 1 with a 5-frame window —
 * run the temporal half's to_qkv projection and its causal temporal attention (model/attention.py:41-71) as ONE kernel.  Outputs
 * are bit-identical to the two-kernel path, which every other shape always uses.  Per eager forward the fused kernel measured 1-2 %
 * slower, per replayed captured step 1-1.6 % FASTER: a harness times both at warm-up and keeps the faster
 * (gtav_amd.generate.tune_weight_prefetch does).  The first enabling call allocates
 * head-major copies of the temporal to_qkv weights (6 D^2 bytes per block) and, on a finalized handle, fills them and
 * synchronises the device; every call drops the captured graphs of the handle.  Ignored on handles with training enabled. */
int gtav_dit_set_fused_temporal(gtav_dit* h, int32_t enable);

/* The spatial counterpart, ON by default on handles whose frames are 144 tokens (hidden_size % 256 == 0): steps / forwards of 5 or more frames (the batch-1
 * window step: 80 (frame, head) blocks; the context-cached step of 5 or more batch items; every batched window step) run the spatial half's to_qkv projection
 * and its attention (model/attention.py:16-38) as ONE kernel; the spatial q / k / v^T never leave the chip.  Bit-identical to the two-kernel path, which
 * every other shape, bf16 half-blocks, training handles and every other geometry always use; measured faster at every eligible size.
 * enable = 0 selects the two-kernel path.  gtav_dit_create allocates the head-major weight copies (6 D^2 bytes per block); every call that changes the
 * setting drops the captured graphs of the handle. */
int gtav_dit_set_fused_spatial(gtav_dit* h, int32_t enable);
/* Which of the two fused launches a forward / sampler step of B x T frames starting at window frame t0 runs on this handle as it is now (switches, operand types,
 * training): bit 0 the spatial one, bit 1 the temporal one (in any block).  NOT a status code.  gtav_dit_profile books a fused launch under the attention class of
 * its half (attn_spatial / attn_temporal) — one class, one kernel — and the to_qkv class then holds the remaining plain to_qkv launches only. */
int gtav_dit_fused_launches(gtav_dit* h, int32_t B, int32_t T, int32_t t0);

/* L2 prefetch of the NEXT GEMM's weight by the small-M GEMM launches (docs/LABNOTES.md 4.10; steps of 256 ... 1536 tokens; default: mode 0x11441).  It changes no
 * arithmetic — results are bit-identical under every setting — and what pays depends on the GPU (profiles/round5/prefetch_box_survey.txt): on some MI355X
 * GPUs prefetching every weight takes 7-12 % off a batch-1 step, on others that costs 1-8 % while the to_qkv / out-proj weights alone or the first K
 * tiles of each weight still gain 1-3 %.  So a harness times a few captured steps per setting and keeps the fastest (gtav_amd.generate.tune_weight_prefetch
 * does; bench.py reports the choice).  mode: 0 = off, 1 = every weight, whole slice; or 0x10000 | one nibble per consumer class — bits 0-3 the
 * out-projection's weight (prefetched by the to_qkv launch), 4-7 fc1's (by the out-projection), 8-11 fc2's (by fc1), 12-15 to_qkv's (by fc2) — with
 * nibble 0 = not prefetched, 1 = the whole slice, k >= 2 = the first k K tiles of every row tile.  Every call that changes the setting drops the captured
 * graphs of the handle. */
int gtav_dit_set_weight_prefetch(gtav_dit* h, int32_t mode);

/* In-situ kernel timing for bench.py's roofline line: when enabled, every kernel of a forward is bracketed by
 * HIP events on the launch stream and the forward synchronises at its end (measurement passes only).
 * Classes: 0 LN+modulate, 1 QKV GEMM, 2 spatial attention, 3 temporal attention, 4 out-proj GEMM, 5 fc1 GEMM,
 * 6 fc2 GEMM, 7 other (patchify, embed, final, unpatchify), 8 an EMPTY event pair (the timing overhead per pair, to be
 * subtracted from every class average).  Conditioning kernels are not included.  A fused to_qkv + attention launch
 * (gtav_dit_fused_launches) is ONE kernel and is booked under the attention class of its half (2 or 3); class 1 then holds
 * the remaining plain to_qkv launches only. */
#define GTAV_PROFILE_CLASSES 9
int gtav_dit_profile(gtav_dit* h, int32_t enable);
int gtav_dit_profile_read(gtav_dit* h, double* ms_by_class, int64_t* launches_by_class);

/* Calibration of the in-situ profiler's timer: `reps` back-to-back launches of a one-wave kernel that spins `spin_us` microseconds on the device's own 100 MHz
 * clock, each with an event pair attached to its dispatch like a profiled kernel.  *event_us_mean = what the event pairs read, *device_us_mean = what the kernel
 * measured itself; the difference is the constant an attached pair adds (bench.py subtracts it, less the dispatch ramp rocprofv3 also counts).  Allocates,
 * creates events and synchronises `stream`. */
int gtav_timer_calibrate(int32_t spin_us, int32_t reps, double* event_us_mean, double* device_us_mean, void* stream);

/* ---- DiT training step ------------------------------------------------------------------------------------
 * Replaces, for the DiT, what train_dit.py does through torch autograd / torch.optim / accelerate:
 *   :649-650 `v_pred = self.dit(x_noisy, t, actions); loss = mse_loss(v_pred[:, -1:], v_target)`  -> gtav_dit_train_forward (+ gtav_mse)
 *   :680     `self.accelerator.backward(scaled_loss)`                                              -> gtav_dit_train_backward
 *   :232-238 `AdamW(self.dit.parameters(), lr, weight_decay, betas=(0.9, 0.999), eps=1e-7)`, :965-970 `clip_grad_norm_`,
 *            `optimizer.step()`, `optimizer.zero_grad()`                                           -> gtav_dit_adamw_step, gtav_dit_zero_grad
 *   DDP's gradient all-reduce (accelerate)                                                         -> one all-reduce over the gradient arena
 *                                                                                                     (gtav_comm_allreduce_f32 or torch.distributed)
 * Mixed precision: fp16 MFMA operands with fp32 accumulation, fp32 master weights / gradients / optimizer state, a loss scale
 * (default 65536) instead of bf16's exponent range; a non-finite gradient norm skips the step (gtav_dit_train_stats reports it).
 * gtav_dit_train_enable must be called right after gtav_dit_create (before any gtav_dit_set_weight).  grad_arena_dev (optional):
 * caller-owned device buffer of gtav_dit_train_param_count floats that receives all gradients contiguously, parameters in the
 * lexicographic order of their state-dict names (NULL: the handle allocates it).  These calls allocate (enable) or synchronise
 * (train_stats); forward / backward / adamw_step only enqueue. */
int gtav_dit_train_param_count(gtav_dit* h, int64_t* numel);
int gtav_dit_train_enable(gtav_dit* h, float* grad_arena_dev, int64_t grad_arena_numel);
int gtav_dit_set_loss_scale(gtav_dit* h, float scale);
/* Data-parallel training: the gradient arena holds the SUM over `divisor` ranks (all-reduce SUM) — gtav_dit_adamw_step and gtav_dit_train_stats
 * then work on arena / (loss scale x divisor), i.e. the rank average DDP would have produced, without a pass over the 2.4 GB arena.  Default 1. */
int gtav_dit_set_grad_divisor(gtav_dit* h, float divisor);
int gtav_dit_zero_grad(gtav_dit* h, void* stream);
/* DiT.forward in training mode: same result as gtav_dit_forward, keeps the activations the backward pass needs. */
int gtav_dit_train_forward(gtav_dit* h, const float* x_dev, const int64_t* t_dev, const float* actions_dev, float* out_dev,
                           int32_t B, int32_t T, void* stream);
/* Adds loss_scale * d mean((v_pred[:, -1] - v_target)^2) / d theta to the gradient arena.  v_pred (B,T,C,H,W) as returned by
 * gtav_dit_train_forward, v_target (B,C,H,W). */
int gtav_dit_train_backward(gtav_dit* h, const float* v_pred_dev, const float* v_target_dev, void* stream);
/* Residual stream of the last training forward: state r_k after k of the 4*depth branch additions (k even: 0 = patch embedding,
 * 4 = output of block 0, ..., 4*depth = input of the final layer), fp32 [B*T*P][hidden] in token order (b, t, p). */
int gtav_dit_train_get_residual(gtav_dit* h, int32_t k, float* dst_dev, int64_t numel, void* stream);
/* The same pass in phases [phase_begin, phase_end): 0 = loss + final layer, p in 1..depth = block depth - p (after it every gradient named
 * "blocks.<depth-p>.*" is complete), depth + 1 = patch embedding + timestep / action embedders.  Lets the host all-reduce one block's
 * slice of the arena (gtav_dit_train_param_range) on another stream while earlier blocks are still being differentiated (the bucketed,
 * overlapped gradient all-reduce DDP performs under accelerate). */
int gtav_dit_train_backward_phases(gtav_dit* h, const float* v_pred_dev, const float* v_target_dev, int32_t phase_begin, int32_t phase_end,
                                   void* stream);
/* Arena slice [offset, offset + count) of the parameters whose state-dict names start with `prefix` (e.g. "blocks.7."). */
int gtav_dit_train_param_range(gtav_dit* h, const char* prefix, int64_t* offset, int64_t* count);
/* Raw (loss-scaled) gradient of one parameter in torch layout. */
int gtav_dit_get_grad(gtav_dit* h, const char* name, float* dst_dev, int64_t numel, void* stream);
int gtav_dit_adamw_step(gtav_dit* h, float lr, float beta1, float beta2, float eps, float weight_decay, float max_grad_norm, void* stream);
/* (gtav_dit_adamw_step: the step is SKIPPED on the device — weights, moments and the Adam step count untouched, "skipped steps" + 1 —
 * when the global gradient norm is not finite or when an fp16 gradient / activation store saturated since the last step: every such
 * store clamps to +-65504 and raises a bit in the handle's error word, which the step consumes.  The Adam step count t and its bias
 * corrections 1 - beta^t are kept on the device and advance with APPLIED steps only.) */
/* out4_host: sum of squares of the scaled gradients, step coefficient (0 = skipped), skipped steps so far, unscaled gradient norm. */
int gtav_dit_train_stats(gtav_dit* h, float* out4_host, void* stream);
/* Optimizer state for checkpoint / resume (train_dit.py:765-849 accelerator.save_state / load_state): the AdamW first / second moments of
 * one parameter (fp32, the parameter's state-dict shape; the fp32 master itself goes through gtav_dit_get_weight / set_weight) and the
 * counters (applied steps = the Adam step count; skipped steps).  get_opt_step / set_opt_step synchronise `stream`. */
int gtav_dit_get_opt_state(gtav_dit* h, const char* name, float* m_dst_dev, float* v_dst_dev, int64_t numel, void* stream);
int gtav_dit_set_opt_state(gtav_dit* h, const char* name, const float* m_src_dev, const float* v_src_dev, int64_t numel, void* stream);
int gtav_dit_get_opt_step(gtav_dit* h, int64_t* applied_steps, int64_t* skipped_steps, void* stream);
int gtav_dit_set_opt_step(gtav_dit* h, int64_t applied_steps, int64_t skipped_steps, void* stream);

/* Reads and clears the handle's device error word (synchronises): fails if, since the last call, a timestep was outside
 * [0, 999], an input held a NaN/inf, or an fp16 activation store saturated (|x| > 65504 is clamped to +-65504, never
 * inf: the reference runs bf16, which has fp32 range; a checkpoint with outlier channels is reported instead of
 * silently producing NaN). */
int gtav_dit_check(gtav_dit* h, void* stream);

/* ---- operand type of the 2-byte tensors (range safety) ---------------------------------------------------------------------------
 * The reference runs this path under bf16 autocast by default (generate.py:125-127 `Accelerator(mixed_precision=...)`, train_dit.py:190-198; train_dit.denoise_step's
 * `dtype=torch.bfloat16`): fp32 exponent range, 8 mantissa bits.  This library defaults to fp16 operands (fp32 accumulation), which is what holds the forward within
 * 1e-3 relative L2 of the fp32 reference — but an activation beyond +-65504 is clamped (and reported by gtav_dit_check).  For checkpoints with such outliers every
 * 2-byte tensor of a group of layers — LayerNorm output, q / k / v, attention output, MLP hidden, temporal K/V cache and the group's GEMM weights — can be bf16 instead:
 * the same kernels and layouts compiled for bf16 operands (v_mfma_f32_16x16x32_bf16), ~8e-3 relative L2 per forward when every group is bf16.
 *   groups of a DiT handle: 2 l = spatial half of block l, 2 l + 1 = its temporal half, 2 depth = the patch embedding, 2 depth + 1 = the final layer
 *   gtav_dit_set_operand_dtype(h, group, dtype)   group -1 = every group.  The fp16 / bf16 images of the changed groups' GEMM weights become stale: send those weights
 *       again with gtav_dit_set_weight (any weight may be re-sent) and call gtav_dit_finalize.  Drops the captured graphs and the cached context.  Training handles
 *       keep fp16 operands.
 *   gtav_dit_autorange(h, &n, stream)   gtav_dit_check that RECOVERS: every fp16 group whose stores saturated since the last check is switched to bf16
 *       (*n_switched of them; weights to be re-sent as above, results since the last check to be recomputed); other errors are reported like gtav_dit_check.
 * The VAE handle has one type for all its layers. */
#define GTAV_OPERAND_F16 0
#define GTAV_OPERAND_BF16 1
int gtav_dit_operand_groups(gtav_dit* h, int32_t* n_groups);
int gtav_dit_set_operand_dtype(gtav_dit* h, int32_t group, int32_t dtype);
int gtav_dit_get_operand_dtype(gtav_dit* h, int32_t group, int32_t* dtype);
int gtav_dit_autorange(gtav_dit* h, int32_t* n_switched, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ViT-VAE — replaces model/vae.py:160-361 (class AutoencoderKL)
 * ---------------------------------------------------------------------------------------------- */
typedef struct gtav_vae_config {
    /* AutoencoderKL.__init__ arguments, model/vae.py:161-176 */
    int32_t latent_dim, input_height, input_width, patch_size;
    int32_t enc_dim, enc_depth, enc_heads, dec_dim, dec_depth, dec_heads;
    float mlp_ratio;
    int32_t use_variational;
    int32_t max_frames_per_call; /* workspace capacity: frames per encode/decode call */
} gtav_vae_config;

typedef struct gtav_vae gtav_vae;

int gtav_vae_create(const gtav_vae_config* cfg, gtav_vae** out);
void gtav_vae_destroy(gtav_vae* h);
int gtav_vae_set_weight(gtav_vae* h, const char* name, const float* src_dev, int64_t numel, void* stream);
int gtav_vae_finalize(gtav_vae* h, void* stream);
int gtav_vae_get_weight(gtav_vae* h, const char* name, float* dst_dev, int64_t numel, void* stream);

/* AutoencoderKL.encode (model/vae.py:306-322): img (N,3,H,W) f32; the network sees in_scale*img + in_shift
 * (callers pass 2,-1 to fuse generate.py:56 `x * 2 - 1`; 1,0 for the plain method).
 * moments_dev (N, seq_len, 2*latent) f32 = quant_conv output with logvar already clamped to [-30, 20]
 * (DiagonalGaussianDistribution, model/vae.py:19-30). */
int gtav_vae_encode(gtav_vae* h, const float* img_dev, float in_scale, float in_shift, float* moments_dev, int32_t N,
                    void* stream);
/* AutoencoderKL.decode (model/vae.py:324-338): z (N, seq_len, latent) f32, network input is z_scale*z;
 * img (N,3,H,W) f32 = out_scale*decoded + out_shift (1,0 for the plain method). */
int gtav_vae_decode(gtav_vae* h, const float* z_dev, float z_scale, float* img_dev, float out_scale, float out_shift,
                    int32_t N, void* stream);

/* Same as gtav_dit_check for the VAE handle (NaN/inf input pixels, fp16 saturation). */
int gtav_vae_check(gtav_vae* h, void* stream);
/* Operand type of every 2-byte tensor of the VAE handle (see gtav_dit_set_operand_dtype): GTAV_OPERAND_F16 (default) or GTAV_OPERAND_BF16.  A change makes every GEMM
 * weight image stale: send the weights again (gtav_vae_set_weight), then gtav_vae_finalize. */
int gtav_vae_set_operand_dtype(gtav_vae* h, int32_t dtype);
int gtav_vae_get_operand_dtype(gtav_vae* h, int32_t* dtype);
/* In-situ kernel timing of encode / decode, as gtav_dit_profile (same GTAV_PROFILE_CLASSES order; class 3, temporal attention, stays empty; 4 = the
 * attention projection, 7 = patchify / patch embedding / quant_conv / post_quant_conv / predictor / unpatchify): bench.py's config4 roofline object.
 * With it enabled every encode / decode call synchronises `stream` at its end. */
int gtav_vae_profile(gtav_vae* h, int32_t enable);
int gtav_vae_profile_read(gtav_vae* h, double* ms_by_class, int64_t* launches_by_class);

/* ------------------------------------------------------------------------------------------------
 * Sampler / training elementwise math on caller-owned buffers
 * ---------------------------------------------------------------------------------------------- */
/* train_dit.py:110-125 for `rows` frames of n elements each, per-row alphas (device arrays). */
int gtav_ddim_update(const float* x_dev, const float* v_dev, float* out_dev, int32_t rows, int32_t n,
                     const float* alpha_t_dev, const float* alpha_next_dev, int32_t is_final, void* stream);
/* train_dit.py:625-641: out = x*sqrt(a) + sqrt(1-a)*clamp(noise, +-clamp_abs), per-row alpha. */
int gtav_add_noise(const float* x_dev, const float* noise_dev, const float* alpha_dev, float* out_dev, int32_t rows,
                   int32_t n, float clamp_abs, void* stream);
/* train_dit.py:643-645: v_target = sqrt(a)*clamp(noise) - sqrt(1-a)*x. */
int gtav_vtarget(const float* x_dev, const float* noise_dev, const float* alpha_dev, float* vt_dev, int32_t rows,
                 int32_t n, float clamp_abs, void* stream);
/* y[i] += alpha * x[i] (x may alias y): the trainer's `total_loss += loss` / `total_loss / n` on device scalars (train_dit.py:676,682). */
int gtav_axpy_f32(float* y_dev, const float* x_dev, float alpha, int64_t n, void* stream);
/* train_dit.py:650 mse_loss: out[0] = mean((a-b)^2) over rows x n; a,b rows are a_stride / b_stride floats apart.
 * out_dev must hold 1 + rows floats. */
int gtav_mse(const float* a_dev, int64_t a_stride, const float* b_dev, int64_t b_stride, int32_t rows, int32_t n,
             float* out_dev, void* stream);
/* generate.py:201-202: x (B, F, n) f32, frames [first, F) of every sample are clamped to [lo, hi] in place
 * (`torch.clamp(chunk, -noise_abs_max, +noise_abs_max)` on the initial noise of the generated frames). */
int gtav_clamp_frames(float* x_dev, int32_t B, int32_t F, int32_t first, int32_t n, float lo, float hi, void* stream);
/* generate.py:238-244 tail: uint8 = clamp(img * 255, 0, 255) with img (N,3,H,W) f32 -> (N,H,W,3) u8. */
int gtav_frames_to_u8(const float* img_dev, uint8_t* out_dev, int32_t N, int32_t H, int32_t W, void* stream);
/* generate.py:56-65: latents (N,C,h,w) = scale * mean, from moments (N, h*w, 2*latent) (first `latent` channels). */
int gtav_moments_to_latents(const float* moments_dev, float* lat_dev, int32_t N, int32_t hw, int32_t latent,
                            int32_t mom_ch, float scale, void* stream);
/* Dataset step (web_dataset.py:41-57,105-107, hf_dataset.py:22-41): a decoded strip image uint8 (H, n_frames*W, 3) — five 270x480 frames
 * side by side in the GTAV dataset — becomes frames (n_frames, 3, OH, OW) f32 in [0,1]: ToTensor (/255, HWC->CHW), SplitImages,
 * Resize((OH,OW)) bilinear with antialiasing (torch F.interpolate(..., antialias=True), what torchvision's tensor Resize calls). */
int gtav_strip_to_frames(const uint8_t* strip_dev, int32_t H, int32_t W, int32_t n_frames, float* out_dev, int32_t OH, int32_t OW,
                         void* stream);
/* generate.py:150-153 prompt path: float frames (N,3,H,W) -> (N,3,OH,OW), same filter. */
int gtav_resize_frames(const float* src_dev, float* dst_dev, int32_t N, int32_t H, int32_t W, int32_t OH, int32_t OW, void* stream);
/* generate.py:238: (N,C,h,w) latents -> (N, h*w, C) decoder input. */
int gtav_latents_to_tokens(const float* lat_dev, float* z_dev, int32_t N, int32_t hw, int32_t latent, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Kernel-level entry points (used by the parity tests in tests/ and by bench.py's roofline probe)
 * ---------------------------------------------------------------------------------------------- */
/* epilogues: 0 f32, 1 f16, 2 gelu-tanh f16, 3 gelu-erf f16, 4 residual (+gate) f32 in place,
 * 6 split-K partial slabs out[ks][M][N] f32 (the split factor is passed in gate_stride), 7 f16 tile-major without an activation (the training forward's
 * MLP pre-activation; rows padded to 128, ldo % 64 == 0) */
int gtav_op_gemm_f16(const void* x_f16_dev, int32_t ldx, const void* w_f16_dev, const float* bias_dev, void* out_dev,
                     int32_t ldo, int32_t M, int32_t N, int32_t K, int32_t epilogue, const float* gate_dev,
                     int32_t gate_stride, int32_t rows_per_gate, void* stream);
/* fused QKV projection + RoPE + attention-layout scatter. mode 0 spatial, 1 temporal (see csrc/gemm.h).
 * rope_cs_dev: interleaved (cos, sin) table [npos][32][2] built by gtav_op_rope_interleave from cos/sin [npos][64]. */
int gtav_op_gemm_qkv(const void* x_f16_dev, int32_t ldx, const void* w_f16_dev, const float* bias_dev, int32_t M,
                     int32_t D, int32_t mode, void* q_dev, void* k_dev, void* v_dev, int32_t S, int32_t Tq, int32_t t0,
                     int32_t Tmax, const float* rope_cs_dev, void* stream);
int gtav_op_rope_interleave(const float* cos_dev, const float* sin_dev, float* cs_dev, int32_t npos, void* stream);
int gtav_op_skinny_f32(const float* x_dev, int32_t ldx, const float* w_dev, const float* bias_dev, float* y_dev,
                       int32_t ldy, int32_t M, int32_t N, int32_t K, int32_t act_silu, void* stream);
int gtav_op_ln_modulate(const float* x_dev, void* out_f16_dev, int32_t M, int32_t D, const float* shift_dev,
                        const float* scale_dev, int32_t mod_stride, int32_t rows_per_mod, void* stream);
int gtav_op_ln_affine(const float* x_dev, void* out_f16_dev, int32_t M, int32_t D, const float* gamma_dev,
                      const float* beta_dev, void* stream);
int gtav_op_attn_spatial(const void* q_dev, const void* k_dev, const void* vt_dev, void* o_dev, int32_t NB,
                         int32_t heads, int32_t S, void* stream);
int gtav_op_attn_temporal(const void* q_dev, const void* kv_dev, void* o_dev, int32_t B, int32_t P, int32_t D,
                          int32_t Tq, int32_t t0, int32_t Tmax, void* stream);
/* Temporal half of a full-window step as ONE kernel (gtav_dit_set_fused_temporal): x rows in (b, 16 positions, frame, position in
 * group) order, w = the to_qkv weight in head-major row order (gtav_op_qkv_head_major of the tile-major [3 D][D] weight); writes K / V
 * of every token to the temporal cache kv [B][Tmax][P][2 D] and the attention output o (f16 tile-major, rows in (b, frame, position)
 * order).  Requires Tq == 5, t0 == 0, P % 16 == 0, (M / 80) * (D / 64) <= 256 (model/attention.py:41-71). */
int gtav_op_qkv_head_major(const void* w_f16_dev, void* w_hm_f16_dev, int32_t D, void* stream);
int gtav_op_gemm_qkvt_attn(const void* x_tperm_f16_dev, const void* w_hm_f16_dev, int32_t M, int32_t D, int32_t P, int32_t Tq,
                           int32_t t0, int32_t Tmax, const float* rope_cs_dev, void* kv_dev, void* o_dev, void* stream);
/* Spatial half of a window step as ONE kernel (gtav_dit_set_fused_spatial): x = the LayerNorm output (f16 tile-major, rows in (b, frame, position) order),
 * w = the to_qkv weight in the wave-interleaved head-major row order (gtav_op_qkv_head_major_spatial of the tile-major [3 D][D] weight), rope_cs = the spatial
 * table [P][64]; writes the attention output o (f16 tile-major) and nothing else — no q / k / v^T leave the chip.  Requires P == 144, D % 64 == 0,
 * (M / 144) * (D / 64) <= 256 (model/attention.py:16-38). */
int gtav_op_qkv_head_major_spatial(const void* w_f16_dev, void* w_hm_f16_dev, int32_t D, void* stream);
int gtav_op_gemm_qkvs_attn(const void* x_f16_dev, const void* w_hm_f16_dev, int32_t M, int32_t D, int32_t P, const float* rope_cs_dev, void* o_dev,
                           void* stream);
/* Backward of the spatial attention (model/attention.py:99-136) for NB x heads (frame, head) items of S tokens: q, k [item][S][64]
 * (RoPE applied), vt [item][64][S], d_o fp16 row-major [NB S][heads 64]; writes the gradient of the to_qkv output, fp16 tile-major
 * logical [NB S][3 heads 64] (dq | dk | dv, dq / dk rotated back through the RoPE).  S % 16 == 0, S <= 160. */
int gtav_op_attn_spatial_bwd(const void* q_dev, const void* k_dev, const void* vt_dev, const void* d_o_dev, int32_t NB, int32_t heads,
                             int32_t S, const float* rope_cs_dev, void* dqkv_dev, void* stream);
/* Weight-gradient GEMM (train_dit.py:680 accelerator.backward, the dW = dY^T X of every Linear): out[m][n] += sum_t x[t][m] * w[t][n] with
 * both operands the ordinary tile-major fp16 activations [K tokens][features] (x: M features, w: N features); out f32 row-major [M][ldo],
 * accumulated in place.  M, N multiples of 128, K a multiple of 64. */
int gtav_op_gemm_tn(const void* x_f16_dev, const void* w_f16_dev, int32_t M, int32_t N, int32_t K, float* out_dev, int32_t ldo, void* stream);
/* The weight-gradient GEMMs of one DiT half-block in ONE launch of 256 x 256 tiles (the training step's dW = dY^T X of fc2, fc1, to_out, to_qkv,
 * train_dit.py:680): for each of the n <= 4 groups out_g[m][n] (f32 row-major [M_g][ldo_g]) += sum_k x_g[m][k] * w_g[n][k]; x_g / w_g tile-major fp16
 * [M_g][K] / [N_g][K] (the TRANSPOSED activations: K = tokens); M_g, N_g multiples of 256, K of 64.  The arrays of pointers / sizes are HOST arrays. */
int gtav_op_gemm_dw_grouped(int32_t n, const void* const* x_f16_dev, const void* const* w_f16_dev, float* const* out_dev, const int32_t* M,
                            const int32_t* N, const int32_t* ldo, int32_t K, void* stream);
/* Residual GEMM as the model runs it: split-K partial slabs (parts: splitk*M*N floats; splitk 0 = heuristic) followed by
 * the LayerNorm kernel that reduces them: resid += gate * (sum parts + bias); out = LN(resid) * (1 + scale + 1e-6) + shift. */
int gtav_op_gemm_splitk_ln(const void* x_f16_dev, int32_t ldx, const void* w_f16_dev, const float* bias_dev, int32_t M,
                           int32_t N, int32_t K, int32_t splitk, float* parts_dev, float* resid_dev, const float* gate_dev,
                           int32_t gate_stride, int32_t rows_per_gate, void* out_f16_dev, const float* shift_dev,
                           const float* scale_dev, int32_t mod_stride, void* stream);
int gtav_op_gemm_choose_splitk(int32_t M, int32_t N, int32_t K);
/* 1 when the model runs a residual GEMM of this shape (out-proj, fc2) with the in-place gated residual epilogue (epilogue 4) instead of split-K slabs
 * reduced by the following LayerNorm: large M on the persistent loader-wave kernel (csrc/gemm.h gemm_resid_inplace_ok). */
int gtav_op_gemm_resid_inplace(int32_t M, int32_t N, int32_t K);
/* (The per-thread hooks the parity tests use to force a GEMM block shape / pipeline depth are declared in gtav_amd_testing.h: they are
 * not part of the product interface.) */
/* fp32 [R][C] -> fp16 [Rp][Cp] zero padded; tiled != 0 writes the GEMM's tile-major operand layout (128 x 64 tiles,
 * csrc/common.h tiled_off; Rp % 128 == 0, Cp % 64 == 0).  All fp16 GEMM operands (x_f16_dev, w_f16_dev) and the fp16
 * outputs of gtav_op_ln_*, gtav_op_attn_* and the GELU epilogues use that layout. */
int gtav_op_convert_f16(const float* src_dev, int32_t lds, int32_t R, int32_t C, void* dst_f16_dev, int32_t Rp,
                        int32_t Cp, int32_t tiled, void* stream);

/* ---- collectives: RCCL over xGMI through the C-ABI, one communicator per process / GPU ------------------
 * Replaces what the reference gets from accelerate / torch.distributed (train_dit.py:199-203 Accelerator, DDP gradient averaging;
 * README.md:119-122 `accelerate launch`, one process per GPU).  Rank 0 calls gtav_comm_unique_id and hands the 128 bytes to the other
 * ranks out of band (MPI, a file, a socket); every rank then calls gtav_comm_init on its own device.  librccl.so is opened at run
 * time (a copy already loaded into the process is reused).  allreduce: in place, fp32, sum or average; allgather: rank r's
 * bytes_per_rank bytes land at recv + r * bytes_per_rank on every rank.  Calls enqueue on `stream` (init / destroy synchronise). */
typedef struct gtav_comm gtav_comm;
int gtav_comm_unique_id(void* id128_host);
int gtav_comm_init(gtav_comm** out, int32_t nranks, int32_t rank, const void* id128_host);
int gtav_comm_allreduce_f32(gtav_comm* c, float* buf_dev, int64_t count, int32_t average, void* stream);
int gtav_comm_allgather(gtav_comm* c, const void* send_dev, void* recv_dev, int64_t bytes_per_rank, void* stream);
int gtav_comm_destroy(gtav_comm* c);

#ifdef __cplusplus
}
#endif
#endif /* GTAV_AMD_H */
