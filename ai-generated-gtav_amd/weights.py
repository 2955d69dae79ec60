"""State-dict layout of the reference models, deterministic synthetic weights, safetensors I/O.

Names and shapes follow the reference's torch state-dict:
  DiT  — model/dit.py:233-293 (x_embedder, t_embedder, external_cond, blocks.N.{s,t}_{attn,mlp,adaLN_modulation},
         final_layer) plus the de-duplicated rotary `freqs` aliases.
  VAE  — model/vae.py:161-236 (patch_embed, encoder/decoder.N.{norm1,attn,norm2,mlp}, enc_norm, quant_conv,
         post_quant_conv, dec_norm, predictor).
This module is host-side plumbing (CPU torch tensors only); it performs no model compute.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, Iterable, Tuple

import torch

Shapes = "OrderedDict[str, Tuple[int, ...]]"


def rope_freqs_pixel(dim: int, max_freq: float) -> torch.Tensor:
    """model/rotary_embedding_torch.py:124-125."""
    return torch.linspace(1.0, max_freq / 2, dim // 2) * math.pi


def rope_freqs_lang(dim: int, theta: float = 10000.0) -> torch.Tensor:
    """model/rotary_embedding_torch.py:120-123."""
    return 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))


def dit_param_shapes(input_h=18, input_w=32, patch_size=2, in_channels=16, hidden_size=1024, depth=12,
                     num_heads=16, mlp_ratio=4.0, external_cond_dim=25, **_) -> Shapes:
    D, p = hidden_size, patch_size
    Hm = int(D * mlp_ratio)
    s: Shapes = OrderedDict()
    s["x_embedder.proj.weight"] = (D, in_channels, p, p)
    s["x_embedder.proj.bias"] = (D,)
    s["t_embedder.mlp.0.weight"] = (D, 256)
    s["t_embedder.mlp.0.bias"] = (D,)
    s["t_embedder.mlp.2.weight"] = (D, D)
    s["t_embedder.mlp.2.bias"] = (D,)
    if external_cond_dim > 0:
        s["external_cond.weight"] = (D, external_cond_dim)
        s["external_cond.bias"] = (D,)
    for i in range(depth):
        for h in ("s", "t"):
            b = f"blocks.{i}.{h}_"
            s[b + "attn.to_qkv.weight"] = (3 * D, D)
            s[b + "attn.to_out.weight"] = (D, D)
            s[b + "attn.to_out.bias"] = (D,)
            s[b + "mlp.fc1.weight"] = (Hm, D)
            s[b + "mlp.fc1.bias"] = (Hm,)
            s[b + "mlp.fc2.weight"] = (D, Hm)
            s[b + "mlp.fc2.bias"] = (D,)
            s[b + "adaLN_modulation.1.weight"] = (6 * D, D)
            s[b + "adaLN_modulation.1.bias"] = (6 * D,)
    s["final_layer.linear.weight"] = (p * p * in_channels, D)
    s["final_layer.linear.bias"] = (p * p * in_channels,)
    s["final_layer.adaLN_modulation.1.weight"] = (2 * D, D)
    s["final_layer.adaLN_modulation.1.bias"] = (2 * D,)
    return s


def dit_freq_alias_names(depth: int) -> Iterable[str]:
    """The 2 + 2*depth aliases of the two shared rotary `freqs` parameters (model/dit.py:259-262)."""
    yield "spatial_rotary_emb.freqs"
    yield "temporal_rotary_emb.freqs"
    for i in range(depth):
        yield f"blocks.{i}.s_attn.rotary_emb.freqs"
        yield f"blocks.{i}.t_attn.rotary_emb.freqs"


def vae_param_shapes(latent_dim=16, input_height=360, input_width=640, patch_size=20, enc_dim=1024, enc_depth=6,
                     enc_heads=16, dec_dim=1024, dec_depth=12, dec_heads=16, mlp_ratio=4.0,
                     use_variational=True, **_) -> Shapes:
    s: Shapes = OrderedDict()
    s["patch_embed.proj.weight"] = (enc_dim, 3, patch_size, patch_size)
    s["patch_embed.proj.bias"] = (enc_dim,)

    def blocks(prefix, depth, dim):
        hm = int(dim * mlp_ratio)
        for i in range(depth):
            b = f"{prefix}.{i}."
            s[b + "norm1.weight"] = (dim,)
            s[b + "norm1.bias"] = (dim,)
            s[b + "attn.qkv.weight"] = (3 * dim, dim)
            s[b + "attn.qkv.bias"] = (3 * dim,)
            s[b + "attn.proj.weight"] = (dim, dim)
            s[b + "attn.proj.bias"] = (dim,)
            s[b + "norm2.weight"] = (dim,)
            s[b + "norm2.bias"] = (dim,)
            s[b + "mlp.fc1.weight"] = (hm, dim)
            s[b + "mlp.fc1.bias"] = (hm,)
            s[b + "mlp.fc2.weight"] = (dim, hm)
            s[b + "mlp.fc2.bias"] = (dim,)

    blocks("encoder", enc_depth, enc_dim)
    s["enc_norm.weight"] = (enc_dim,)
    s["enc_norm.bias"] = (enc_dim,)
    mult = 2 if use_variational else 1
    s["quant_conv.weight"] = (mult * latent_dim, enc_dim)
    s["quant_conv.bias"] = (mult * latent_dim,)
    s["post_quant_conv.weight"] = (dec_dim, latent_dim)
    s["post_quant_conv.bias"] = (dec_dim,)
    blocks("decoder", dec_depth, dec_dim)
    s["dec_norm.weight"] = (dec_dim,)
    s["dec_norm.bias"] = (dec_dim,)
    s["predictor.weight"] = (3 * patch_size ** 2, dec_dim)
    s["predictor.bias"] = (3 * patch_size ** 2,)
    return s


# ---------------------------------------------------------------------------------------------
# deterministic synthetic weights (no checkpoints exist offline; SURVEY.md §8(d) "Synthetic inputs")
# ---------------------------------------------------------------------------------------------
def _std_for(name: str, shape: Tuple[int, ...]) -> Tuple[float, float]:
    """(mean, std) of the synthetic value of a parameter.  Chosen so that — unlike the reference's own
    init, where every adaLN gate is zero and each block is the identity (model/dit.py:316-320) —
    every kernel's output matters in the final result: O(1) modulation/gates, O(1) residual updates."""
    if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("_norm.weight"):
        return 1.0, 0.1
    if name.endswith(".bias"):
        return 0.0, 0.02
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    if "adaLN_modulation" in name:
        return 0.0, 1.0 / math.sqrt(fan_in)          # shift/scale/gate ~ O(|silu(c)|)
    if name.startswith("t_embedder.mlp.0"):
        return 0.0, 2.0 / math.sqrt(fan_in)
    if name.startswith("t_embedder.mlp.2") or name.startswith("external_cond"):
        return 0.0, 2.0 / math.sqrt(fan_in)
    if name.startswith("final_layer.linear"):
        return 0.0, 1.0 / math.sqrt(fan_in)
    return 0.0, 1.0 / math.sqrt(fan_in)              # variance-preserving linear maps


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    """Hash-seeded N(mean, std) tensor: identical on every machine with this torch build."""
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    mean, std = _std_for(name, shape)
    return torch.randn(shape, generator=g, dtype=torch.float32) * std + mean


def synth_state_dict(shapes: Shapes, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((n, synth_tensor(n, s, seed)) for n, s in shapes.items())


# ---------------------------------------------------------------------------------------------
# safetensors I/O with the reference's alias conventions (SURVEY.md §8(b) "Weight layout")
# ---------------------------------------------------------------------------------------------
def load_state_dict_file(path: str) -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file
    return dict(load_file(path, device="cpu"))


def save_state_dict_file(sd: Dict[str, torch.Tensor], path: str) -> None:
    from safetensors.torch import save_file
    save_file({k: v.detach().cpu().contiguous() for k, v in sd.items()}, path)


def split_freq_keys(sd: Dict[str, torch.Tensor]):
    """Separates rotary `freqs` entries (any alias; they are non-learned constants,
    rotary_embedding_torch.py:136) from real parameters. Returns (params, spatial_freqs|None, temporal_freqs|None)."""
    params, sf, tf = {}, None, None
    for k, v in sd.items():
        if k.endswith("rotary_emb.freqs"):
            if k.startswith("spatial_") or ".s_attn." in k:
                sf = v
            else:
                tf = v
        else:
            params[k] = v
    return params, sf, tf
