"""Batch-generic generation harness (reference generate.py:50-66,186-244): VAE-encode the prompt frames,
autoregressive denoising with a sliding window, VAE-decode.  The per-frame initial noise is an input
(CPU-generated, so CPU-oracle and GPU runs see identical noise) and the batch can be sharded across
ranks (one process per GPU) with a final all-gather of the latents."""
from __future__ import annotations

from typing import Optional

import time

import torch

from . import lib as _lib
from .utils import alphas_cumprod as _alphas_cumprod

SCALING_FACTOR = 0.07843137255  # generate.py:50


@torch.inference_mode()
def vae_encode(x: torch.Tensor, vae, n_prompt_frames: int, scaling_factor: float = SCALING_FACTOR) -> torch.Tensor:
    """generate.py:50-66: frames (B,t,3,H,W) in [0,1] -> latents (B,t,C,h,w) = vae.encode(2x-1).mean * s."""
    B, t = x.shape[:2]
    H, W = x.shape[-2:]
    mom = vae.encode_moments(x.reshape(B * t, *x.shape[2:]), 2.0, -1.0)
    N, hw, mom_ch = mom.shape
    lat = torch.empty((N, vae.latent_dim, hw), device=mom.device, dtype=torch.float32)
    with torch.cuda.device(mom.device):
        _lib.check(_lib.load().gtav_moments_to_latents(mom.data_ptr(), lat.data_ptr(), N, hw, vae.latent_dim, mom_ch,
                                                       scaling_factor, _lib.current_stream()))
    return lat.reshape(B, t, vae.latent_dim, H // vae.patch_size, W // vae.patch_size)


@torch.inference_mode()
def vae_decode_frames(x: torch.Tensor, vae, scaling_factor: float = SCALING_FACTOR, to_uint8: bool = True) -> torch.Tensor:
    """generate.py:238-244: latents (B,t,C,h,w) -> uint8 frames (B,t,H,W,3) (or float (B,t,3,H,W) in [0,1]-ish)."""
    B, t, Cc, h, w = x.shape
    xd = x.to(vae.device, torch.float32).contiguous()
    z = torch.empty((B * t, h * w, Cc), device=xd.device, dtype=torch.float32)
    L = _lib.load()
    with torch.cuda.device(xd.device):
        _lib.check(L.gtav_latents_to_tokens(xd.data_ptr(), z.data_ptr(), B * t, h * w, Cc, _lib.current_stream()))
    img = vae.decode(z, 1.0 / scaling_factor, 0.5, 0.5)          # (decode(x / s) + 1) / 2
    if not to_uint8:
        return img.reshape(B, t, 3, vae.input_height, vae.input_width)
    out = torch.empty((B * t, vae.input_height, vae.input_width, 3), device=img.device, dtype=torch.uint8)
    with torch.cuda.device(img.device):
        _lib.check(L.gtav_frames_to_u8(img.data_ptr(), out.data_ptr(), B * t, vae.input_height, vae.input_width,
                                       _lib.current_stream()))
    return out.reshape(B, t, vae.input_height, vae.input_width, 3)


@torch.inference_mode()
def generate_latents(model, x_prompt: torch.Tensor, total_frames: int, noise_steps: int, noise_chunks: torch.Tensor,
                     actions: Optional[torch.Tensor] = None, stabilization_level: int = 15, noise_abs_max: float = 20.0,
                     clamp_min: float = 1e-4, ctx_cache: bool = False, hoist_cond: bool = True) -> torch.Tensor:
    """generate.py:186-220.  x_prompt (B, n_prompt, C, h, w) latents; noise_chunks (B, total-n_prompt, C, h, w)
    standard-normal draws (clamped to +-noise_abs_max here, generate.py:201-202); actions (B, total, 25) or None.
    ctx_cache=False re-runs the whole window on every noise step exactly like the reference;
    ctx_cache=True runs the window once per generated frame and then only the frame being denoised,
    taking the context K/V of the temporal layers from the cache (same result, ~4.8x less work).
    hoist_cond=True builds the conditioning (adaLN) table of all noise steps of a frame in one batch per frame (the
    conditioning never depends on x); False recomputes it inside every step like DiT.forward does. Same results.
    Returns latents (B, total_frames, C, h, w) on the model's device."""
    dev = model.device
    B, n_prompt = x_prompt.shape[:2]
    x = torch.empty((B, total_frames, *x_prompt.shape[2:]), device=dev, dtype=torch.float32)
    x[:, :n_prompt] = x_prompt.to(dev, torch.float32)                          # copies only: torch is storage here
    x[:, n_prompt:] = noise_chunks.to(dev, torch.float32)
    fsz = x[0, 0].numel()
    with torch.cuda.device(dev):                                               # generate.py:201-202 clamp, as a HIP kernel
        _lib.check(_lib.load().gtav_clamp_frames(x.data_ptr(), B, total_frames, n_prompt, fsz, -float(noise_abs_max),
                                                 float(noise_abs_max), _lib.current_stream()))
    act = actions.to(dev, torch.float32).contiguous() if actions is not None else None
    model.set_schedule(_alphas_cumprod(clamp_min))
    noise_range = torch.linspace(0, 999, noise_steps + 1)                      # generate.py:194 (float)
    t_of = [int(v) for v in noise_range]                                       # long() truncation (train_dit.py:70)
    order = list(reversed(range(0, noise_steps + 1)))
    x_init = x.clone() if getattr(model, "range_policy", "report") == "auto" else None
    for attempt in range(1 + (model.n_operand_groups if x_init is not None else 0)):
        for i in range(n_prompt, total_frames):
            start = max(0, i + 1 - model.max_frames)                              # generate.py:204
            if hoist_cond:
                model.prepare_frame_(B, total_frames, start, i, stabilization_level, [t_of[k] for k in order], act)
            for step, noise_idx in enumerate(order):
                cached = ctx_cache and noise_idx != noise_steps
                model.denoise_step_(x, start, i, stabilization_level, t_of[noise_idx], t_of[max(0, noise_idx - 1)],
                                    noise_idx <= 0, act, cached=cached, cond_step=step if hoist_cond else -1)
        try:
            model.check()
            break
        except _lib.GtavRangeSwitch:
            # range_policy "auto": the layers whose fp16 stores saturated run on bf16 operands from here on; the clip is clipped — generate it again from
            # the same inputs (each retry can only add layer groups, so the loop ends)
            x.copy_(x_init)
    return x


PREFETCH_CLASSES = ("out", "fc1", "fc2", "qkv")      # consumer classes of gtav_dit_set_weight_prefetch's per-class mode, nibble order


def prefetch_mode(cls) -> int:
    """(out, fc1, fc2, qkv) values (0 = that weight is not prefetched, 1 = its whole slice, k >= 2 = the first k K tiles of every row tile) -> the mode
    word of gtav_dit_set_weight_prefetch (include/gtav_amd.h)."""
    if not any(cls):
        return 0
    if all(v == 1 for v in cls):
        return 1
    return 0x10000 | cls[0] | cls[1] << 4 | cls[2] << 8 | cls[3] << 12


@torch.inference_mode()
def tune_weight_prefetch(model, B: int = 1, window: Optional[int] = None, steps: int = 24, rounds: int = 2, use_actions: bool = False,
                         latent_hw=None, per_class: bool = True, fused_temporal: bool = True, fused_spatial: bool = True) -> dict:
    """Times the captured full-window sampler step of batch B under different settings of the next-weight L2 prefetch (docs/LABNOTES.md 4.10) and leaves
    the model on the fastest.  The results are bit-identical under every setting, only the speed differs, and what pays depends on the GPU: on some
    MI355X GPUs prefetching every weight takes 7-12 % off the step; on the others that costs 1-8 % (the prefetched lines are gone before their consumer
    starts and the weight is read twice), but the to_qkv / out-proj weights alone, or the first K tiles of each weight, still take 1-3 % off
    (profiles/round5/prefetch_box_survey.txt).  First off, on for every weight and the library's default against each other (`rounds` alternations, `steps` replays each, a fresh
    capture per switch); then, with `per_class`, from the fastest of them one pass over the four weight classes (fc2, fc1, out-proj, to_qkv) trying skip / first 4 K tiles / whole
    slice for each while the others stay.  About 0.5 s + 2.5 s at batch 1; synthetic latents.
    With `fused_temporal` / `fused_spatial` the fused to_qkv + attention launches are timed the same way afterwards (eligible windows only) and kept where faster.
    Returns {"on_ms", "off_ms", "chosen": "on" | "off" | "per-class", "classes": {...}, "mode", "tuned_ms", "fused_temporal_qkv_attention": {...} | None,
    "fused_spatial_qkv_attention": {...} | None}
    (milliseconds per step)."""
    dev = model.device
    T = int(window or model.max_frames)
    h, w = latent_hw or (model.input_h, model.input_w)
    F = T
    x0 = (torch.randn(B, F, model.in_channels, h, w, generator=torch.Generator().manual_seed(11)) * 0.5).to(dev)
    act = None
    if use_actions:
        act = torch.zeros(B, F, 25, device=dev)
        act[:, :, 3] = 1
    model.set_schedule(_alphas_cumprod(1e-4))
    ts = [999 - 7 * k for k in range(steps + 3)]

    def timed(cls) -> float:
        model.set_weight_prefetch(prefetch_mode(cls))
        x = x0.clone()
        model.prepare_frame_(B, F, 0, T - 1, 15, ts, act)
        for k in range(3):                                  # eager warm-up, capture, first replay
            model.denoise_step_(x, 0, T - 1, 15, ts[k], ts[k + 1], False, act, cond_step=k)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(3, steps + 3):
            model.denoise_step_(x, 0, T - 1, 15, ts[k], ts[min(k + 1, steps + 2)], False, act, cond_step=k)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / steps * 1e3

    on, off, safe = (1, 1, 1, 1), (0, 0, 0, 0), (1, 4, 4, 1)      # safe: the library's default (never slower than off on any GPU surveyed)
    starts = (on, off, safe) if per_class else (on, off)
    best = {cls: float("inf") for cls in starts}
    for _ in range(rounds):
        for cls in starts:
            best[cls] = min(best[cls], timed(cls))
    cur = min(starts, key=lambda cls: best[cls])
    cur_ms = best[cur]
    if per_class:
        for c in (2, 1, 0, 3):
            for v in (0, 4, 1):
                if v == cur[c]:
                    continue
                cand = cur[:c] + (v,) + cur[c + 1:]
                ms = min(timed(cand) for _ in range(rounds))
                if ms < cur_ms * 0.997:                     # a change has to show: 0.3 % is the repeatability of this timing
                    cur, cur_ms = cand, ms
    model.set_weight_prefetch(prefetch_mode(cur))
    # The temporal to_qkv + temporal attention as ONE launch (gtav_dit_set_fused_temporal: bit-identical to the two-kernel path, five-frame windows of at most 1 280
    # tokens): slower per eager forward when it was built (round 2), but in the captured step a launch less per temporal half-block is worth more than the kernel
    # loses on some GPUs (round 6: 2.044 -> 2.011 ms per step) — so it is timed here like the prefetch and kept only where it is faster.
    fused = None
    if fused_temporal and T == 5 and B * T * (h // model.patch_size) * (w // model.patch_size) <= 1280 and not getattr(model, "_trainable", False):
        was = bool(getattr(model, "_fused_temporal", False))
        t_off = t_on = float("inf")
        for _ in range(rounds):
            model.set_fused_temporal(False)
            t_off = min(t_off, timed(cur))
            model.set_fused_temporal(True)
            t_on = min(t_on, timed(cur))
        keep = t_on < t_off * 0.997
        model.set_fused_temporal(keep)
        timed(cur)                                            # leaves the handle warmed up (and its step captured) under the kept setting
        fused = {"on_ms": round(t_on, 4), "off_ms": round(t_off, 4), "chosen": "on" if keep else "off", "was": was}
        cur_ms = min(cur_ms, t_on) if keep else cur_ms
    # The spatial counterpart (gtav_dit_set_fused_spatial: frames of 144 tokens, 5 or more frames per step; the library's default where eligible), timed the same
    # way on top of whatever the temporal switch was left at.
    fused_s = None
    if fused_spatial and (h // model.patch_size) * (w // model.patch_size) == 144 and B * T >= 5 and model.hidden_size % 256 == 0 and not getattr(model, "_trainable", False):
        was = getattr(model, "_fused_spatial", None)
        was = True if was is None else bool(was)              # None: the library's default, on for this geometry
        t_off = t_on = float("inf")
        for _ in range(rounds):
            model.set_fused_spatial(False)
            t_off = min(t_off, timed(cur))
            model.set_fused_spatial(True)
            t_on = min(t_on, timed(cur))
        keep = t_on < t_off * 0.997
        model.set_fused_spatial(keep)
        timed(cur)
        fused_s = {"on_ms": round(t_on, 4), "off_ms": round(t_off, 4), "chosen": "on" if keep else "off", "was": was}
        cur_ms = min(cur_ms, t_on) if keep else cur_ms
    model.check()
    chosen = "on" if cur == on else "off" if cur == off else "per-class"
    return {"on_ms": round(best[on], 4), "off_ms": round(best[off], 4), "chosen": chosen, "classes": dict(zip(PREFETCH_CLASSES, cur)),
            "mode": prefetch_mode(cur), "tuned_ms": round(cur_ms, 4), "fused_temporal_qkv_attention": fused, "fused_spatial_qkv_attention": fused_s}


def sample_inputs(gid: int, n_prompt: int, total_frames: int, frame_hw, latent_hw, latent_ch: int = 16, seed: int = 1000):
    """Synthetic inputs of ONE sequence, a function of its GLOBAL sample id only (SURVEY.md §8(d),(e)): prompt frames
    U[0,1) (n_prompt, 3, H, W) and the standard-normal initial noise of every generated frame (total - n_prompt, C, h, w),
    drawn from a CPU generator seeded with seed + gid — never device RNG, so CPU-oracle and GPU runs, and any sharding
    of the batch over ranks, see identical draws."""
    g = torch.Generator().manual_seed(int(seed) + int(gid))
    frames = torch.rand(n_prompt, 3, *frame_hw, generator=g)
    noise = torch.randn(total_frames - n_prompt, latent_ch, *latent_hw, generator=g)
    return frames, noise


def shard_inputs(global_batch: int, rank: int, world: int, n_prompt: int, total_frames: int, frame_hw, latent_hw,
                 latent_ch: int = 16, seed: int = 1000):
    """Inputs of rank `rank`'s contiguous shard of a global batch (one process per GPU): returns
    (global_ids, frames (b, n_prompt, 3, H, W), noise (b, total - n_prompt, C, h, w)) on the CPU.  Concatenating the
    shards of all ranks in rank order gives exactly the world-size-1 batch (tests/test_host_cpu.py proves it under gloo)."""
    lo, hi = shard_batch(global_batch, rank, world)
    ins = [sample_inputs(g, n_prompt, total_frames, frame_hw, latent_hw, latent_ch, seed) for g in range(lo, hi)]
    return list(range(lo, hi)), torch.stack([f for f, _ in ins]), torch.stack([n for _, n in ins])


def generate_clip(dit, vae, frames: torch.Tensor, noise: torch.Tensor, total_frames: int, noise_steps: int,
                  actions: Optional[torch.Tensor] = None, ctx_cache: bool = False, to_uint8: bool = True, gather: bool = True):
    """One rank's whole clip (reference generate.py:main for its shard of the batch): VAE-encode the prompt frames,
    run the denoising loop, all-gather the final latents over the ranks (the path's only collective, RCCL over xGMI
    under the nccl backend) and decode this rank's own frames.  Returns (gathered latents, decoded local frames)."""
    n_prompt = frames.shape[1]
    x0 = vae_encode(frames.to(vae.device), vae, n_prompt)
    x = generate_latents(dit, x0, total_frames, noise_steps, noise, actions, ctx_cache=ctx_cache)
    xg = all_gather_latents(x) if gather else x
    out = vae_decode_frames(x, vae, to_uint8=to_uint8)
    vae.check()
    return xg, out


def shard_batch(B: int, rank: int, world: int):
    """Contiguous batch shard of rank `rank`: samples [lo, hi). Requires B % world == 0."""
    assert B % world == 0, f"global batch {B} must be divisible by world size {world}"
    per = B // world
    return rank * per, (rank + 1) * per


def all_gather_latents(x_local: torch.Tensor) -> torch.Tensor:
    """One all-gather (RCCL over xGMI when the backend is nccl) of the per-rank latents along the batch dim."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return x_local
    out = torch.empty((dist.get_world_size() * x_local.shape[0], *x_local.shape[1:]), device=x_local.device, dtype=x_local.dtype)
    dist.all_gather_into_tensor(out, x_local.contiguous())
    return out
