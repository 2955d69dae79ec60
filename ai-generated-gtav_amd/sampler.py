"""Sampler math on the GPU (reference train_dit.py:30-125 `denoise_step`)."""
from __future__ import annotations

from typing import Optional

import torch

from . import lib as _lib


def ddim_update(x: torch.Tensor, v: torch.Tensor, alpha_t: torch.Tensor, alpha_next: Optional[torch.Tensor], is_final: bool):
    """train_dit.py:110-125 on (rows, n)-shaped fp32 device tensors with per-row alphas."""
    rows = alpha_t.numel()
    n = x.numel() // rows
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().gtav_ddim_update(x.data_ptr(), v.data_ptr(), out.data_ptr(), rows, n, alpha_t.data_ptr(),
                                                _lib.ptr(alpha_next), int(bool(is_final)), _lib.current_stream()))
    return out


@torch.inference_mode()
def denoise_step(dit_model, x_noisy, actions, noise_idx, stabilization_level, noise_range, alphas_cumprod, start_frame=0,
                 dtype=None):
    """Same signature and return value as train_dit.denoise_step (train_dit.py:30-125): returns
    (x_pred, v_pred) for the window x_noisy[:, start_frame:].  `dtype` is the reference's autocast type (train_dit.py:105-107): torch.bfloat16 /
    torch.float16 select the model's operand type (DiT.set_operand_dtype: accumulation, residual stream, LayerNorm, softmax and the conditioning path are
    fp32 either way).  One deviation: the DEFAULT is None = leave the model as it is (fp16 operands unless it was switched), where the reference's default is
    torch.bfloat16 — fp16 is what keeps a forward within 1e-3 of the fp32 reference; pass torch.bfloat16 for the reference's own range / precision."""
    if dtype is not None:
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError(f"denoise_step: dtype {dtype} (torch.float16, torch.bfloat16 or None)")
        dit_model.set_operand_dtype(dtype)
    dev = dit_model.device
    B = x_noisy.shape[0]
    t_ctx = torch.full((B, x_noisy.shape[1] - 1), int(stabilization_level), dtype=torch.long)
    t = torch.full((B, 1), int(noise_range[noise_idx]), dtype=torch.long)               # long() truncation, :70-76
    t_next = torch.full((B, 1), int(noise_range[max(0, noise_idx - 1)]), dtype=torch.long)
    t = torch.cat([t_ctx, t], dim=1)[:, start_frame:]
    t_next = torch.cat([t_ctx, t_next], dim=1)[:, start_frame:]
    x_curr = x_noisy[:, start_frame:].to(dev, torch.float32).contiguous()
    if actions is not None:
        actions = actions[:, start_frame: start_frame + x_curr.shape[1]]
    v_pred = dit_model(x_curr, t, actions)
    ac = alphas_cumprod.reshape(-1).float().cpu()
    alpha_t = ac[t.reshape(-1)].to(dev).contiguous()
    alpha_next = ac[t_next].clone()
    alpha_next[:, :-1] = 1.0                                                            # train_dit.py:117-118
    alpha_next = alpha_next.reshape(-1).to(dev).contiguous()
    x_pred = ddim_update(x_curr, v_pred, alpha_t, alpha_next, noise_idx <= 0)
    return x_pred, v_pred
