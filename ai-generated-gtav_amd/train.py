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


@torch.inference_mode()
def forward_loss(dit, latents: torch.Tensor, actions: Optional[torch.Tensor], target_noise_idx: torch.Tensor,
                 ctx_noise_idx: torch.Tensor, ctx_noise: torch.Tensor, noise: torch.Tensor, noise_steps: int = 50,
                 n_prompt_frames: int = 4, noise_abs_max: float = 20.0, clamp_min: float = 1e-6, keep_activations: bool = False):
    """train_dit.py:590-650 for clips of n_prompt_frames + 1 frames. Returns (loss (1,) tensor, v_pred, v_target).
    keep_activations: run the DiT through forward_train so that dit.backward_(v_pred, v_target) can follow."""
    dev = dit.device
    L = _lib.load()
    B, total = latents.shape[:2]
    assert total == n_prompt_frames + 1
    nr = torch.linspace(0, 999, noise_steps + 1).long()                                 # train_dit.py:309-315
    ac = _alphas_cumprod(clamp_min)
    i = n_prompt_frames
    ctx_noise_idx = torch.minimum(ctx_noise_idx.cpu(), target_noise_idx.cpu())       # train_dit.py:587
    start = max(0, i + 1 - dit.max_frames)
    t = torch.zeros((B, i + 1), dtype=torch.long)
    t[:, :-1] = nr[ctx_noise_idx].unsqueeze(1)
    t[:, -1] = nr[target_noise_idx.cpu()]
    t = t[:, start:]
    W = t.shape[1]
    x_curr = latents[:, start: i + 1].to(dev, torch.float32).contiguous()
    a = actions[:, start: i + 1].to(dev, torch.float32).contiguous() if actions is not None else None
    n = x_curr[0, 0].numel()
    alpha = ac[t].to(dev).contiguous()                                                  # (B, W)
    all_noise = torch.cat([ctx_noise.to(dev, torch.float32), noise.to(dev, torch.float32)], dim=1).contiguous()
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
    """Data-parallel gradient averaging (what DDP does under accelerate): one all-reduce over the contiguous arena, in place."""
    if world_size > 1:
        import torch.distributed as dist
        dist.all_reduce(dit.grad_arena, op=dist.ReduceOp.SUM)
        dit.grad_arena.div_(world_size)


def gradient_buckets(dit):
    """The all-reduce buckets of the overlapped backward, in the order their gradients become final: (phase after which the bucket is
    complete, arena offset, count).  One bucket per block (38 M floats = 151 MB for DiT-S/2: large enough for RCCL's ring over xGMI to run
    at link rate, small enough that 15 of the 16 hide behind the blocks still being differentiated), the final layer after phase 0, the
    embedders last.  The arena is laid out in lexicographic name order, so "blocks.<l>." is one contiguous slice."""
    L = dit.depth
    buckets = [(0,) + dit.param_range("final_layer.")]
    buckets += [(L - l,) + dit.param_range(f"blocks.{l}.") for l in reversed(range(L))]
    rest = [dit.param_range(p) for p in ("external_cond.", "t_embedder.", "x_embedder.")]
    buckets += [(L + 1,) + r for r in rest]
    return buckets


def backward_overlapped(dit, v_pred, v_target, world_size: int, comm_stream=None, all_reduce=None):
    """backward_ in phases with the all-reduce of each finished bucket enqueued on `comm_stream` while the compute stream differentiates
    the next block (what DDP's bucketed reducer does under accelerate).  `all_reduce(tensor)` defaults to
    torch.distributed.all_reduce (SUM) and may be gtav_amd.comm.Comm.all_reduce_.  Gradients are averaged (divided by world_size)."""
    import torch.distributed as dist
    if all_reduce is None:
        all_reduce = lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM)
    comm_stream = comm_stream or torch.cuda.Stream(device=dit.device)
    arena = dit.grad_arena
    buckets = gradient_buckets(dit)
    done = 0
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
            for _, off, cnt in ready:
                all_reduce(arena[off: off + cnt])
                done += cnt
    if world_size > 1:
        torch.cuda.current_stream(dit.device).wait_stream(comm_stream)
        assert done == arena.numel(), "gradient buckets must cover the arena"
        arena.div_(world_size)


@torch.inference_mode()
def training_step(dit, latents: torch.Tensor, actions: Optional[torch.Tensor], target_noise_idx: torch.Tensor, ctx_noise_idx: torch.Tensor,
                  ctx_noise: torch.Tensor, noise: torch.Tensor, lr: float, weight_decay: float = 0.0, max_grad_norm: float = 1.0,
                  world_size: int = 1, noise_steps: int = 50, n_prompt_frames: int = 4, noise_abs_max: float = 20.0, clamp_min: float = 1e-6,
                  overlap_all_reduce: bool = True, comm_stream=None, all_reduce=None):
    """One optimisation step on a batch of (n_prompt_frames + 1)-frame latent clips: forward + loss (train_dit.py:590-650), backward,
    gradient all-reduce (bucketed and overlapped with the backward pass when world_size > 1), clip, AdamW.  Returns the loss tensor (1,)."""
    dit.zero_grad()
    loss, v_pred, v_target = forward_loss(dit, latents, actions, target_noise_idx, ctx_noise_idx, ctx_noise, noise, noise_steps, n_prompt_frames,
                                          noise_abs_max, clamp_min, keep_activations=True)
    if world_size > 1 and overlap_all_reduce:
        backward_overlapped(dit, v_pred, v_target, world_size, comm_stream, all_reduce)
    else:
        dit.backward_(v_pred, v_target)
        all_reduce_gradients(dit, world_size)
    dit.adamw_step(lr, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
    return loss
