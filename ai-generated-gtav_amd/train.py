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
                 n_prompt_frames: int = 4, noise_abs_max: float = 20.0, clamp_min: float = 1e-6):
    """train_dit.py:590-650 for clips of n_prompt_frames + 1 frames. Returns (loss (1,) tensor, v_pred, v_target)."""
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
    v_pred = dit(x_noisy, t, a)
    out = torch.empty(1 + B, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        vp_last = v_pred[:, -1]
        _lib.check(L.gtav_mse(vp_last.data_ptr(), v_pred.stride(0), v_target.data_ptr(), n, B, n, out.data_ptr(), stream))
    return out[:1], v_pred, v_target.reshape(B, 1, *x_curr.shape[2:])
