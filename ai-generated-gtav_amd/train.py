"""Training forward + loss (reference train_dit.py:329-351 `encode_frames`, :554-682 `_shared_step`, forward part):
noise the context and target frames, run the DiT once over the window, MSE against the v-target of the
last frame.  Random draws are arguments so that parity runs can inject them."""
from __future__ import annotations

from typing import Optional

import torch

from . import lib as _lib
from .generate import vae_encode
from .utils import alphas_cumprod as _alphas_cumprod


@torch.inference_mode()
def encode_frames(vae, frames: torch.Tensor) -> torch.Tensor:
    """train_dit.py:329-351."""
    return vae_encode(frames, vae, frames.shape[1])


def _frame_step(dit, latents, actions, i, target_noise_idx, ctx_noise_idx, ctx_noise, noise, nr, ac, noise_abs_max, keep_activations):
    """One iteration of the frame loop of `_shared_step` (train_dit.py:590-650) for target frame i: returns (loss (1,), v_pred, v_target)."""
    dev = dit.device
    L = _lib.load()
    B = latents.shape[0]
    tgt = [int(v) for v in target_noise_idx]
    ctx = [min(int(c), t_) for c, t_ in zip(ctx_noise_idx, tgt)]                       # train_dit.py:587 torch.minimum
    start = max(0, i + 1 - dit.max_frames)
    t = torch.zeros((B, i + 1), dtype=torch.long)
    t[:, :-1] = nr[torch.tensor(ctx)].unsqueeze(1)
    t[:, -1] = nr[torch.tensor(tgt)]
    t = t[:, start:]
    W = t.shape[1]
    x_curr = latents[:, start: i + 1].to(dev, torch.float32).contiguous()
    a = actions[:, start: i + 1].to(dev, torch.float32).contiguous() if actions is not None else None
    n = x_curr[0, 0].numel()
    alpha = ac[t].to(dev).contiguous()                                                  # (B, W)
    all_noise = torch.empty_like(x_curr)                                                # copies only: torch is storage here
    all_noise[:, :-1] = ctx_noise.to(dev, torch.float32)
    all_noise[:, -1:] = noise.to(dev, torch.float32)
    x_noisy = torch.empty_like(x_curr)
    stream = _lib.current_stream()
    with torch.cuda.device(dev):
        _lib.check(L.gtav_add_noise(x_curr.data_ptr(), all_noise.data_ptr(), alpha.data_ptr(), x_noisy.data_ptr(), B * W, n,
                                    noise_abs_max, stream))
        x_last = x_curr[:, -1].contiguous()
        nz_last = all_noise[:, -1].contiguous()
        a_last = alpha[:, -1].contiguous()
        v_target = torch.empty_like(x_last)
        _lib.check(L.gtav_vtarget(x_last.data_ptr(), nz_last.data_ptr(), a_last.data_ptr(), v_target.data_ptr(), B, n,
                                  noise_abs_max, stream))
    v_pred = dit.forward_train(x_noisy, t, a) if keep_activations else dit(x_noisy, t, a)
    out = torch.empty(1 + B, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        vp_last = v_pred[:, -1]
        _lib.check(L.gtav_mse(vp_last.data_ptr(), v_pred.stride(0), v_target.data_ptr(), n, B, n, out.data_ptr(), stream))
    return out[:1], v_pred, v_target.reshape(B, 1, *x_curr.shape[2:])


def _per_frame(arg, n_iter, what):
    """A per-iteration argument of the frame loop: a list / tuple of n_iter entries, or (one target frame) the entry itself."""
    if isinstance(arg, (list, tuple)):
        assert len(arg) == n_iter, f"{what}: {len(arg)} entries for {n_iter} target frames"
        return list(arg)
    if torch.is_tensor(arg) and what.endswith("idx") and arg.dim() == 2:
        assert arg.shape[0] == n_iter, f"{what}: {arg.shape[0]} rows for {n_iter} target frames"
        return [arg[k] for k in range(n_iter)]
    assert n_iter == 1, f"{what}: clips with {n_iter} target frames need one entry per target frame"
    return [arg]


@torch.inference_mode()
def forward_loss(dit, latents: torch.Tensor, actions: Optional[torch.Tensor], target_noise_idx, ctx_noise_idx, ctx_noise, noise,
                 noise_steps: int = 50, n_prompt_frames: int = 4, noise_abs_max: float = 20.0, clamp_min: float = 1e-6,
                 keep_activations: bool = False, on_frame=None):
    """train_dit.py:590-682 `_shared_step`: for every target frame i in [n_prompt_frames, total_frames) noise the window that ends at i,
    run the DiT over it, MSE against the v-target of frame i; returns (mean loss over the target frames (1,), v_pred, v_target) of the LAST
    target frame.  The shipped dataset has ONE target frame (5-frame clips): then the draws are plain tensors — target_noise_idx /
    ctx_noise_idx (B,), ctx_noise (B, W - 1, C, h, w), noise (B, 1, C, h, w).  Longer clips pass one entry per target frame (lists, or
    (n, B) index tensors); the window of target frame i holds min(i + 1, max_frames) frames.
    keep_activations: run the DiT through forward_train so that dit.backward_(v_pred, v_target) can follow; on_frame(k, v_pred, v_target)
    is called after target frame k's forward (the trainer differentiates each frame's loss right there, train_dit.py:679-680)."""
    B, total = latents.shape[:2]
    n_iter = total - n_prompt_frames
    assert n_iter >= 1, "forward_loss: no target frame behind the prompt frames"
    tgts, ctxs = _per_frame(target_noise_idx, n_iter, "target_noise_idx"), _per_frame(ctx_noise_idx, n_iter, "ctx_noise_idx")
    cns, nzs = _per_frame(ctx_noise, n_iter, "ctx_noise"), _per_frame(noise, n_iter, "noise")
    nr = torch.linspace(0, 999, noise_steps + 1).long()                                 # train_dit.py:309-315
    ac = _alphas_cumprod(clamp_min)
    total_loss = None
    for k, i in enumerate(range(n_prompt_frames, total)):
        loss, v_pred, v_target = _frame_step(dit, latents, actions, i, tgts[k].cpu().tolist(), ctxs[k].cpu().tolist(), cns[k], nzs[k], nr, ac,
                                             noise_abs_max, keep_activations)
        if on_frame is not None:
            on_frame(k, v_pred, v_target)
        if total_loss is None:
            total_loss = loss
        else:
            with torch.cuda.device(dit.device):                                         # total_loss += loss (train_dit.py:676), as a HIP kernel
                _lib.check(_lib.load().gtav_axpy_f32(total_loss.data_ptr(), loss.data_ptr(), 1.0, 1, _lib.current_stream()))
    if n_iter > 1:
        with torch.cuda.device(dit.device):                                             # / (total_frames - n_prompt_frames) (train_dit.py:682)
            _lib.check(_lib.load().gtav_axpy_f32(total_loss.data_ptr(), total_loss.data_ptr(), 1.0 / n_iter - 1.0, 1, _lib.current_stream()))
    return total_loss, v_pred, v_target


# ------------------------------------------------------------------------------------------------------------------------
# Inference helpers of the trainer (reference train_dit.py:352-552): decode_frames, predict, predict_noise.  Same kernels as the
# generation harness; the trainer's constants differ from generate.py's: schedule clamp_min 1e-6, the
# stabilisation level is noise_range[1] of the TRAINING range (19 for ddim_noise_steps = 50), inference range as .long().
# Random draws are arguments (the reference draws them from the global RNG), so parity runs inject identical noise.
# ------------------------------------------------------------------------------------------------------------------------
def stabilization_level(ddim_noise_steps: int = 50) -> int:
    """train_dit.py:309-327: noise_range = linspace(0, 999, steps + 1).long(); stabilization_level = noise_range[1]."""
    return int(torch.linspace(0, 999, ddim_noise_steps + 1).long()[1])


@torch.inference_mode()
def decode_frames(vae, latents: torch.Tensor) -> torch.Tensor:
    """train_dit.py:352-368: latents (B, t, C, h, w) -> uint8 video (B, t, H, W, 3)."""
    from .generate import vae_decode_frames
    return vae_decode_frames(latents, vae, to_uint8=True)


@torch.inference_mode()
def predict(dit, vae, frames: torch.Tensor, actions: Optional[torch.Tensor], new_frame_noise: torch.Tensor, num_frames: int = 32,
            n_prompt_frames: int = 4, ddim_noise_steps: int = 50, ddim_noise_steps_inference: int = 50, noise_abs_max: float = 20.0,
            decode: bool = True):
    """train_dit.py:370-466 `predict`: first sample of the batch, prompt = its first n_prompt_frames, actions padded with "W"
    (index 3) up to num_frames, autoregressive generation with the trainer's constants.  new_frame_noise (1, num_frames -
    n_prompt_frames, C, h, w): the standard-normal draw of every generated frame.  Returns (latents (1, num_frames, C, h, w), uint8
    video (1, num_frames, H, W, 3) or None)."""
    from .generate import generate_latents
    frames = frames[:1, :n_prompt_frames]
    act = None
    if actions is not None:
        act = actions[:1].to(torch.float32)
        if act.shape[1] < num_frames:
            pad = torch.zeros((1, num_frames - act.shape[1], act.shape[2]), dtype=act.dtype, device=act.device)
            pad[:, :, 3] = 1                                                  # drive straight (train_dit.py:386-390)
            act = torch.cat([act, pad], dim=1)
    x = encode_frames(vae, frames.to(vae.device))
    lat = generate_latents(dit, x, num_frames, ddim_noise_steps_inference, new_frame_noise, act,
                           stabilization_level=stabilization_level(ddim_noise_steps), noise_abs_max=noise_abs_max, clamp_min=1e-6)
    return lat, (decode_frames(vae, lat) if decode else None)


@torch.inference_mode()
def predict_noise(dit, vae, frames: torch.Tensor, actions: Optional[torch.Tensor], ctx_noise: torch.Tensor, new_frame_noise: torch.Tensor,
                  ddim_noise_steps: int = 50, ddim_noise_steps_inference: int = 50, noise_abs_max: float = 20.0):
    """train_dit.py:468-552 `predict_noise`: encode every frame of the first clip, noise the context frames at level
    stabilization_level - 1 (train_dit.py:498), replace the last frame by clamped noise and denoise it.
    ctx_noise (1, n - 1, C, h, w), new_frame_noise (1, 1, C, h, w).  Returns (latents, x_noisy before denoising, x after)."""
    from .generate import generate_latents
    dev = dit.device
    L = _lib.load()
    latents = encode_frames(vae, frames[:1].to(vae.device)).to(dev)
    n = latents.shape[1]
    lvl = stabilization_level(ddim_noise_steps)
    ac = _alphas_cumprod(1e-6)
    ctx = latents[:, :-1].contiguous()
    x_ctx = torch.empty_like(ctx)
    alpha = ac[lvl - 1].reshape(1).repeat(n - 1).to(dev).contiguous()
    nz = ctx_noise.to(dev, torch.float32).contiguous()
    fsz = ctx[0, 0].numel()
    with torch.cuda.device(dev):
        _lib.check(L.gtav_add_noise(ctx.data_ptr(), nz.data_ptr(), alpha.data_ptr(), x_ctx.data_ptr(), n - 1, fsz, float(noise_abs_max),
                                    _lib.current_stream()))
    act = actions[:1].to(torch.float32) if actions is not None else None
    x = generate_latents(dit, x_ctx, n, ddim_noise_steps_inference, new_frame_noise, act, stabilization_level=lvl,
                         noise_abs_max=noise_abs_max, clamp_min=1e-6)
    x_old = torch.cat([x_ctx, new_frame_noise.to(dev, torch.float32)], dim=1).contiguous()      # the sequence before denoising
    with torch.cuda.device(dev):
        _lib.check(L.gtav_clamp_frames(x_old.data_ptr(), 1, n, n - 1, fsz, -float(noise_abs_max), float(noise_abs_max), _lib.current_stream()))
    return latents, x_old, x


# ------------------------------------------------------------------------------------------------------------------------
# Optimisation step.  Reference: train_dit.py:680 accelerator.backward, :232-260 AdamW + cosine schedule with
# warm-up and a floor, :965-970 clip_grad_norm_ / optimizer.step / scheduler.step / zero_grad; accelerate's DDP averages the
# gradients of the ranks.  Here: backward kernels into one contiguous gradient arena, ONE all-reduce over it (RCCL over xGMI:
# 2.4 GB of fp32 for DiT-S/2), clipping + AdamW fused on the GPU.
# ------------------------------------------------------------------------------------------------------------------------
def cosine_with_min_lr(step: int, base_lr: float, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.25,
                       min_lr: float = 0.0) -> float:
    """transformers.get_cosine_with_min_lr_schedule_with_warmup as called at train_dit.py:253-260 (num_cycles=0.25): linear warm-up,
    then base_lr * (min_rate + (1 - min_rate) * 0.5 (1 + cos(2 pi cycles progress))), min_rate = min_lr / base_lr."""
    import math
    if step < num_warmup_steps:
        return base_lr * step / max(1, num_warmup_steps)
    progress = (step - num_warmup_steps) / max(1, num_training_steps - num_warmup_steps)
    factor = 0.5 * (1.0 + math.cos(math.pi * num_cycles * 2.0 * progress))
    rate = min_lr / base_lr
    return base_lr * max(0.0, factor * (1.0 - rate) + rate)


def all_reduce_gradients(dit, world_size: int):
    """Data-parallel gradient averaging (what DDP does under accelerate): ONE all-reduce (SUM) over the contiguous arena; the division by
    the world size is not a pass over the 2.4 GB arena but a factor of the optimizer's step coefficient (dit.grad_divisor ->
    gtav_dit_set_grad_divisor): arena / (loss scale x world) is the rank average."""
    if world_size > 1:
        import torch.distributed as dist
        dist.all_reduce(dit.grad_arena, op=dist.ReduceOp.SUM)
    dit.grad_divisor = float(max(1, world_size))


_BUCKET_PREFIXES = ("external_cond.", "t_embedder.", "x_embedder.")


def gradient_buckets(dit):
    """The all-reduce buckets of the overlapped backward, in the order their gradients become final: (phase after which the bucket is
    complete, arena offset, count).  One bucket per block (38 M floats = 151 MB for DiT-S/2: large enough for RCCL's ring over xGMI to run
    at link rate, small enough that 15 of the 16 hide behind the blocks still being differentiated), the final layer after phase 0, the
    embedders last.  The arena is laid out in lexicographic name order, so "blocks.<l>." is one contiguous slice.  Built and VALIDATED once per
    model (cached on it): the buckets must be disjoint and cover the arena — a model with parameters outside these prefixes fails here, before
    any collective is issued; prefixes the model does not have (external_cond_dim = 0) are skipped."""
    cached = getattr(dit, "_gradient_buckets", None)
    if cached is not None and cached[0] == dit.grad_arena.numel():
        return cached[1]
    L = dit.depth
    buckets = [(0,) + dit.param_range("final_layer.")]
    buckets += [(L - l,) + dit.param_range(f"blocks.{l}.") for l in reversed(range(L))]
    buckets += [(L + 1,) + dit.param_range(p) for p in _BUCKET_PREFIXES]
    buckets = [b for b in buckets if b[2] > 0]
    spans = sorted((off, off + cnt) for _, off, cnt in buckets)
    ok = spans and spans[0][0] == 0 and spans[-1][1] == dit.grad_arena.numel() and all(x[1] == y[0] for x, y in zip(spans, spans[1:]))
    if not ok:
        raise RuntimeError("gradient_buckets: the per-block / embedder slices do not tile the gradient arena (a parameter outside the known name "
                           f"prefixes?): spans {spans[:3]} ... {spans[-2:]}, arena {dit.grad_arena.numel()}")
    dit._gradient_buckets = (dit.grad_arena.numel(), buckets)
    return buckets


def backward_overlapped(dit, v_pred, v_target, world_size: int, comm_stream=None, all_reduce=None, bucket_timings=None):
    """backward_ in phases with the all-reduce (SUM) of each finished bucket enqueued on `comm_stream` while the compute stream differentiates
    the next block (what DDP's bucketed reducer does under accelerate).  `all_reduce(tensor)` defaults to
    torch.distributed.all_reduce (SUM) and may be gtav_amd.comm.Comm.all_reduce_.  The arena then holds the SUM over the ranks and
    dit.grad_divisor = world_size tells the optimizer (no pass over the arena).  `bucket_timings` (a list, measurement passes only) receives one
    (phase, bytes, start event, stop event) per bucket, recorded on the communication stream around its all-reduce."""
    if all_reduce is None:
        import torch.distributed as dist
        all_reduce = lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if comm_stream is None:
        comm_stream = getattr(dit, "_comm_stream", None)
        if comm_stream is None:
            comm_stream = dit._comm_stream = torch.cuda.Stream(device=dit.device)       # one per model, not one per step
    arena = dit.grad_arena
    buckets = gradient_buckets(dit)                                                     # validated: covers the arena
    L = dit.depth
    for phase in range(L + 2):
        dit.backward_phases_(v_pred, v_target, phase, phase + 1)
        ready = [b for b in buckets if b[0] == phase]
        if not ready or world_size == 1:
            continue
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dit.device))
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ev)
            for ph, off, cnt in ready:
                if bucket_timings is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(comm_stream)
                all_reduce(arena[off: off + cnt])
                if bucket_timings is not None:
                    e1.record(comm_stream)
                    bucket_timings.append((ph, cnt * 4, e0, e1))
    if world_size > 1:
        torch.cuda.current_stream(dit.device).wait_stream(comm_stream)
    dit.grad_divisor = float(max(1, world_size))


@torch.inference_mode()
def training_step(dit, latents: torch.Tensor, actions: Optional[torch.Tensor], target_noise_idx, ctx_noise_idx, ctx_noise, noise, lr: float,
                  weight_decay: float = 0.0, max_grad_norm: float = 1.0, world_size: int = 1, noise_steps: int = 50, n_prompt_frames: int = 4,
                  noise_abs_max: float = 20.0, clamp_min: float = 1e-6, overlap_all_reduce: bool = True, comm_stream=None, all_reduce=None,
                  bucket_timings=None):
    """One optimisation step on a batch of latent clips: for every target frame forward + loss and its backward (train_dit.py:590-680: each
    frame's loss is differentiated inside the frame loop, the gradients add up), gradient all-reduce (bucketed and overlapped with the LAST
    frame's backward pass when world_size > 1), clip, AdamW.  Returns the mean loss over the target frames, a (1,) tensor."""
    dit.zero_grad()
    n_iter = latents.shape[1] - n_prompt_frames
    overlap = world_size > 1 and overlap_all_reduce

    def on_frame(k, v_pred, v_target):
        if overlap and k == n_iter - 1:
            backward_overlapped(dit, v_pred, v_target, world_size, comm_stream, all_reduce, bucket_timings)
        else:
            dit.backward_(v_pred, v_target)

    loss, _, _ = forward_loss(dit, latents, actions, target_noise_idx, ctx_noise_idx, ctx_noise, noise, noise_steps, n_prompt_frames,
                              noise_abs_max, clamp_min, keep_activations=True, on_frame=on_frame)
    if not overlap:
        all_reduce_gradients(dit, world_size)
    dit.adamw_step(lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
    return loss


class LossScaler:
    """Dynamic loss scale for the fp16 backward pass (the reference trains under bf16 autocast and needs none).  The optimizer step skips
    itself on the device when a gradient overflowed (non-finite norm or a saturated fp16 store: gtav_dit_adamw_step); every `check_every`
    steps this reads the skip counter (one synchronisation) and halves the scale if steps were skipped since the last check, or doubles it
    after `growth_interval` clean steps — torch.cuda.amp.GradScaler's policy at a coarser cadence."""

    def __init__(self, dit, check_every: int = 50, growth_interval: int = 2000, min_scale: float = 1.0, max_scale: float = 2.0 ** 24):
        self.dit, self.check_every, self.growth_interval = dit, int(check_every), int(growth_interval)
        self.min_scale, self.max_scale = float(min_scale), float(max_scale)
        self._calls = 0
        self._clean = 0
        # the counter's value when this scaler starts watching (a resumed run restores it from the checkpoint); a handle that is not built yet
        # gives it on the first update instead
        try:
            self._skipped_seen = int(dit.train_stats()[1]) if getattr(dit, "_handle", None) else None
        except Exception:
            self._skipped_seen = None

    def update(self) -> float:
        """Call once per optimisation step (after adamw_step).  Returns the loss scale the NEXT step will use."""
        self._calls += 1
        if self._calls % self.check_every:
            return self.dit.loss_scale
        _, skipped, _ = self.dit.train_stats()
        if self._skipped_seen is None:
            # first look at the counter: a resumed run restores it from the checkpoint (set_opt_step) — steps skipped BEFORE this scaler existed
            # say nothing about the restored loss scale
            self._skipped_seen = skipped
            return self.dit.loss_scale
        if skipped > self._skipped_seen:
            self.dit.loss_scale = max(self.min_scale, self.dit.loss_scale * 0.5)
            self._clean = 0
        else:
            self._clean += self.check_every
            if self._clean >= self.growth_interval:
                self.dit.loss_scale = min(self.max_scale, self.dit.loss_scale * 2.0)
                self._clean = 0
        self._skipped_seen = skipped
        return self.dit.loss_scale


# ------------------------------------------------------------------------------------------------------------------------
# Checkpoint / resume (reference train_dit.py:746-849): `save_model` = the weights alone as one .safetensors with the reference's key
# names (train_dit.py:758-762); `save_checkpoint` = accelerator.save_state (model + optimizer) + step.json {step, epoch}; `load_checkpoint`
# = load_state + step.json, then the caller skips `steps_in_epoch * gradient_accumulation_steps` batches (train_dit.py:841-843).
# ------------------------------------------------------------------------------------------------------------------------
def save_model(dit, path: str):
    """train_dit.py:746-763: weights only, safetensors, reference key names (incl. the de-duplicated rotary `freqs`)."""
    from . import weights as _w
    if getattr(dit, "_trainable", False) and dit._handle:
        dit.pull_weights()
    _w.save_state_dict_file(dit.state_dict(), path)


def save_state(dit, ckpt_dir: str, global_step: int, epoch: int, extra: Optional[dict] = None):
    """train_dit.py:765-800 `save_checkpoint`: <dir>/model.safetensors (fp32 masters), <dir>/optimizer.safetensors (AdamW moments + step
    counters), <dir>/step.json {"step", "epoch", "loss_scale", ...}.  Rank 0 writes; the caller barriers around it like the reference."""
    import json
    import os
    from safetensors.torch import save_file
    os.makedirs(ckpt_dir, exist_ok=True)
    save_model(dit, os.path.join(ckpt_dir, "model.safetensors"))
    save_file({k: v.contiguous() for k, v in dit.opt_state_dict().items()}, os.path.join(ckpt_dir, "optimizer.safetensors"))
    state = {"step": int(global_step), "epoch": int(epoch), "loss_scale": float(dit.loss_scale)}
    state.update(extra or {})
    with open(os.path.join(ckpt_dir, "step.json"), "w") as f:
        json.dump(state, f)


def load_state(dit, ckpt_dir: str, steps_per_epoch: Optional[int] = None, gradient_accumulation_steps: int = 1) -> dict:
    """train_dit.py:802-849 `load_checkpoint`: restores weights, AdamW state and the loss scale into `dit` (trainable=True) and returns
    step.json's dict plus "skip_iter" = (step % steps_per_epoch) * gradient_accumulation_steps, the number of batches of the current
    epoch the resumed loop has to skip (train_dit.py:841-843), when steps_per_epoch is given."""
    import json
    import os
    from safetensors.torch import load_file
    from . import weights as _w
    dit.load_state_dict(_w.load_state_dict_file(os.path.join(ckpt_dir, "model.safetensors")))
    dit.load_opt_state_dict(load_file(os.path.join(ckpt_dir, "optimizer.safetensors")))
    with open(os.path.join(ckpt_dir, "step.json")) as f:
        state = json.load(f)
    if "loss_scale" in state:
        dit.loss_scale = float(state["loss_scale"])
    if steps_per_epoch:
        state["skip_iter"] = (state["step"] % int(steps_per_epoch)) * int(gradient_accumulation_steps)
    return state
