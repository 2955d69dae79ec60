"""CPU ORACLE (test infrastructure, NOT product code).

A from-scratch fp32 restatement, in plain torch-CPU functional ops, of the one hot path of
ikergarcia1996/AI-Generated-GTAV that this repo accelerates: the spatio-temporal DiT forward,
the ViT-VAE encode/decode, the DDIM-style `denoise_step`, the sigmoid schedule, the autoregressive
sampling loop and the training forward+loss.  Every function cites the reference file:line whose
behaviour it restates.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this module; the product (`ai-generated-gtav_amd/`) never does and fails loudly when its HIP
library is missing.

PARITY PIN: the reference ships no tests and no golden vectors.  This oracle is pinned
against the reference ITSELF, imported on CPU in the build container through `tools/ref_shim.py`:
`tools/make_golden.py` writes `tests/golden/*.safetensors` (inputs, weights, reference outputs) and
`tests/test_oracle_golden.py` checks this file against them (<= 2e-5 rel-L2, fp32 vs fp32).
Third-party arithmetic outside /root/reference (timm `Mlp`, torch SDPA/LayerNorm/GELU) is unpinned
upstream; the golden vectors above are the only pins.

All weights are taken from a flat `{name: tensor}` dict with the reference's state-dict names
(SURVEY.md §8(b)).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SCALING_FACTOR = 0.07843137255  # generate.py:50, train_dit.py:332


# ----------------------------------------------------------------------------------------------
# configs (constructor arguments of the reference classes)
# ----------------------------------------------------------------------------------------------
@dataclass
class DiTConfig:
    """model/dit.py:233-244 ctor args; the `DiT-S/2` factory (dit.py:379-389) uses depth=16."""
    input_h: int = 18
    input_w: int = 32
    patch_size: int = 2
    in_channels: int = 16
    hidden_size: int = 1024
    depth: int = 12
    num_heads: int = 16
    mlp_ratio: float = 4.0
    external_cond_dim: int = 25
    max_frames: int = 5

    @property
    def grid(self):
        return self.input_h // self.patch_size, self.input_w // self.patch_size

    @property
    def head_dim(self):
        return self.hidden_size // self.num_heads


@dataclass
class VAEConfig:
    """model/vae.py:161-176 ctor args; factory `vit-l-20-shallow-encoder` at vae.py:363-380."""
    latent_dim: int = 16
    input_height: int = 360
    input_width: int = 640
    patch_size: int = 20
    enc_dim: int = 1024
    enc_depth: int = 6
    enc_heads: int = 16
    dec_dim: int = 1024
    dec_depth: int = 12
    dec_heads: int = 16
    mlp_ratio: float = 4.0

    @property
    def seq_h(self):
        return self.input_height // self.patch_size

    @property
    def seq_w(self):
        return self.input_width // self.patch_size

    @property
    def seq_len(self):
        return self.seq_h * self.seq_w

    @property
    def patch_dim(self):
        return 3 * self.patch_size ** 2


def dit_s_2() -> DiTConfig:
    return DiTConfig(depth=16)


def vit_l_20_shallow_encoder() -> VAEConfig:
    return VAEConfig()


# ----------------------------------------------------------------------------------------------
# schedule + embeddings
# ----------------------------------------------------------------------------------------------
def sigmoid_beta_schedule(timesteps: int, start=-3, end=3, tau=1.0, clamp_min=1e-4) -> Tensor:
    """utils.py:30-48. float64 sigmoid alpha-bar, rescaled to [clamp_min, 1]; returns betas (f64)."""
    steps = timesteps + 1
    t = torch.linspace(0, timesteps, steps, dtype=torch.float64) / timesteps
    v_start = torch.tensor(start / tau).sigmoid()
    v_end = torch.tensor(end / tau).sigmoid()
    ac = (-((t * (end - start) + start) / tau).sigmoid() + v_end) / (v_end - v_start)
    ac = ac / ac[0]
    ac = ac * (1 - clamp_min) + clamp_min
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.clip(betas, 0, 0.999)


def alphas_cumprod_table(clamp_min=1e-4, max_noise_level=1000) -> Tensor:
    """generate.py:192-198 / train_dit.py:292-297: betas.float() -> 1-b -> cumprod in fp32. (1000,)"""
    betas = sigmoid_beta_schedule(max_noise_level, clamp_min=clamp_min).float()
    return torch.cumprod(1.0 - betas, dim=0)


def noise_range_generate(noise_steps: int, max_noise_level=1000) -> Tensor:
    """generate.py:194: FLOAT linspace; denoise_step truncates it with torch.full(dtype=long)."""
    return torch.linspace(0, max_noise_level - 1, noise_steps + 1)


def noise_range_train(noise_steps: int, max_noise_level=1000) -> Tensor:
    """train_dit.py:309-315: linspace(...).long()."""
    return torch.linspace(0, max_noise_level - 1, noise_steps + 1).long()


def timestep_embedding(t: Tensor, dim=256, max_period=10000) -> Tensor:
    """model/dit.py:96-118. [cos(t f) | sin(t f)], f_k = exp(-ln(max_period) k / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


# ----------------------------------------------------------------------------------------------
# rotary embeddings (model/rotary_embedding_torch.py)
# ----------------------------------------------------------------------------------------------
def rope_freqs_pixel(dim: int, max_freq: float) -> Tensor:
    """rotary_embedding_torch.py:124-125: linspace(1, max_freq/2, dim//2) * pi."""
    return torch.linspace(1.0, max_freq / 2, dim // 2) * math.pi


def rope_freqs_lang(dim: int, theta=10000.0) -> Tensor:
    """rotary_embedding_torch.py:120-123: 1/theta^(2k/dim)."""
    return 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))


def _angles(pos: Tensor, freqs: Tensor) -> Tensor:
    """rotary_embedding_torch.py:334-337: outer product, each freq repeated twice (interleaved)."""
    a = pos.to(freqs.dtype)[:, None] * freqs[None, :]
    return a.repeat_interleave(2, dim=-1)


def rope_angles_axial(h: int, w: int, freqs: Tensor) -> Tensor:
    """rotary_embedding_torch.py:290-317 get_axial_freqs(h, w) for pixel freqs -> (h, w, 2*2*len(freqs)).

    First half of the last dim rotates by the row position, second half by the column position;
    positions are linspace(-1, 1, n)."""
    ah = _angles(torch.linspace(-1, 1, steps=h), freqs)  # (h, 2F)
    aw = _angles(torch.linspace(-1, 1, steps=w), freqs)  # (w, 2F)
    ah = ah[:, None, :].expand(h, w, ah.shape[-1])
    aw = aw[None, :, :].expand(h, w, aw.shape[-1])
    return torch.cat([ah, aw], dim=-1)


def rope_angles_temporal(T: int, freqs: Tensor) -> Tensor:
    """rotary_embedding_torch.py:186-209 rotate_queries_or_keys: positions 0..T-1 -> (T, 2*len(freqs))."""
    return _angles(torch.arange(T, dtype=torch.float32), freqs)


def rotate_half(x: Tensor) -> Tensor:
    """rotary_embedding_torch.py:39-43: interleaved pairs (x0,x1) -> (-x1, x0)."""
    x = x.reshape(*x.shape[:-1], x.shape[-1] // 2, 2)
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).reshape(*x.shape[:-2], -1)


def apply_rope(angles: Tensor, t: Tensor) -> Tensor:
    """rotary_embedding_torch.py:46-73 with start_index=0, scale=1: rotates the first
    angles.shape[-1] features of t, leaves the rest."""
    rot = angles.shape[-1]
    mid, right = t[..., :rot], t[..., rot:]
    mid = mid * angles.cos() + rotate_half(mid) * angles.sin()
    return torch.cat((mid, right), dim=-1)


# ----------------------------------------------------------------------------------------------
# DiT (model/dit.py, model/attention.py)
# ----------------------------------------------------------------------------------------------
def modulate(x: Tensor, shift: Tensor, scale: Tensor) -> Tensor:
    """model/dit.py:19-27. x (B,T,H,W,D); shift/scale (B,T,D). NB the `+1e-6` on scale."""
    scale = scale + 1e-6
    return x * (1 + scale[:, :, None, None, :]) + shift[:, :, None, None, :]


def gate(x: Tensor, g: Tensor) -> Tensor:
    """model/dit.py:30-35."""
    return g[:, :, None, None, :] * x


def _ln(x: Tensor, weight=None, bias=None) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps=1e-6)


def patch_embed(x: Tensor, w: Tensor, b: Tensor, p: int) -> Tensor:
    """model/dit.py:38-76 Conv2d(k=s=p) -> (N, h, w, D)."""
    y = F.conv2d(x, w, b, stride=p)
    return y.permute(0, 2, 3, 1)


def spatial_attention(sd, pre: str, x: Tensor, heads: int, angles: Tensor) -> Tensor:
    """model/attention.py:99-136. x (B,T,H,W,D); full attention over H*W per (b,t)."""
    B, T, H, W, D = x.shape
    d = D // heads
    qkv = F.linear(x, sd[pre + "to_qkv.weight"])
    q, k, v = qkv.chunk(3, dim=-1)

    def split(z):  # B T H W (h d) -> (B T) h H W d
        return z.reshape(B * T, H, W, heads, d).permute(0, 3, 1, 2, 4)

    q, k, v = split(q), split(k), split(v)
    q, k = apply_rope(angles, q), apply_rope(angles, k)
    q, k, v = (z.reshape(B * T, heads, H * W, d).contiguous() for z in (q, k, v))
    o = F.scaled_dot_product_attention(q, k, v, is_causal=False)
    o = o.reshape(B, T, heads, H, W, d).permute(0, 1, 3, 4, 2, 5).reshape(B, T, H, W, D)
    return F.linear(o, sd[pre + "to_out.weight"], sd[pre + "to_out.bias"])


def temporal_attention(sd, pre: str, x: Tensor, heads: int, temporal_freqs: Tensor) -> Tensor:
    """model/attention.py:41-71. causal attention over T per (b,h,w); RoPE positions 0..T-1."""
    B, T, H, W, D = x.shape
    d = D // heads
    qkv = F.linear(x, sd[pre + "to_qkv.weight"])
    q, k, v = qkv.chunk(3, dim=-1)

    def split(z):  # B T H W (h d) -> (B H W) h T d
        return z.reshape(B, T, H, W, heads, d).permute(0, 2, 3, 4, 1, 5).reshape(B * H * W, heads, T, d)

    q, k, v = split(q), split(k), split(v)
    ang = rope_angles_temporal(T, temporal_freqs)
    q, k = apply_rope(ang, q), apply_rope(ang, k)
    q, k, v = (z.contiguous() for z in (q, k, v))
    o = F.scaled_dot_product_attention(q, k, v, is_causal=True)
    o = o.reshape(B, H, W, heads, T, d).permute(0, 4, 1, 2, 3, 5).reshape(B, T, H, W, D)
    return F.linear(o, sd[pre + "to_out.weight"], sd[pre + "to_out.bias"])


def mlp(sd, pre: str, x: Tensor, approximate: str) -> Tensor:
    """timm Mlp (third party; call sites model/dit.py:171-176,190-195, model/vae.py:147-152)."""
    h = F.linear(x, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"])
    h = F.gelu(h, approximate=approximate)
    return F.linear(h, sd[pre + "fc2.weight"], sd[pre + "fc2.bias"])


def dit_cond(sd, cfg: DiTConfig, t: Tensor, external_cond: Optional[Tensor]) -> Tensor:
    """model/dit.py:359-364: c = t_embedder(t) (+ external_cond Linear). t (B,T) -> (B,T,D)."""
    B, T = t.shape
    e = timestep_embedding(t.reshape(-1), 256)
    h = F.silu(F.linear(e, sd["t_embedder.mlp.0.weight"], sd["t_embedder.mlp.0.bias"]))
    c = F.linear(h, sd["t_embedder.mlp.2.weight"], sd["t_embedder.mlp.2.bias"]).reshape(B, T, -1)
    if torch.is_tensor(external_cond):
        c = c + F.linear(external_cond, sd["external_cond.weight"], sd["external_cond.bias"])
    return c


def dit_block(sd, cfg: DiTConfig, i: int, x: Tensor, c: Tensor, s_angles: Tensor, t_freqs: Tensor) -> Tensor:
    """model/dit.py:200-225 SpatioTemporalDiTBlock.forward."""
    p = f"blocks.{i}."
    sc = F.silu(c)
    for half in ("s", "t"):
        m = F.linear(sc, sd[p + f"{half}_adaLN_modulation.1.weight"], sd[p + f"{half}_adaLN_modulation.1.bias"])
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = m.chunk(6, dim=-1)
        xin = modulate(_ln(x), shift_msa, scale_msa)
        if half == "s":
            a = spatial_attention(sd, p + "s_attn.", xin, cfg.num_heads, s_angles)
        else:
            a = temporal_attention(sd, p + "t_attn.", xin, cfg.num_heads, t_freqs)
        x = x + gate(a, gate_msa)
        x = x + gate(mlp(sd, p + f"{half}_mlp.", modulate(_ln(x), shift_mlp, scale_mlp), "tanh"), gate_mlp)
    return x


def dit_unpatchify(x: Tensor, cfg: DiTConfig) -> Tensor:
    """model/dit.py:328-341: (N,h,w,p*p*C) ordered (p,q,c) -> (N,C,h*p,w*p)."""
    N, h, w, _ = x.shape
    p, c = cfg.patch_size, cfg.in_channels
    x = x.reshape(N, h, w, p, p, c)
    x = torch.einsum("nhwpqc->nchpwq", x)
    return x.reshape(N, c, h * p, w * p)


def dit_forward(sd: Dict[str, Tensor], cfg: DiTConfig, x: Tensor, t: Tensor,
                external_cond: Optional[Tensor] = None, taps: Optional[dict] = None) -> Tensor:
    """model/dit.py:343-376 DiT.forward. x (B,T,C,H,W) f32, t (B,T) int64 -> (B,T,C,H,W) f32.
    `taps`, if given, receives the residual stream after each block (for per-layer parity)."""
    B, T, C, H, W = x.shape
    hd = cfg.head_dim
    s_freqs = sd.get("spatial_rotary_emb.freqs", rope_freqs_pixel(hd // 2, 256))
    t_freqs = sd.get("temporal_rotary_emb.freqs", rope_freqs_lang(hd))
    gh, gw = cfg.grid
    s_angles = rope_angles_axial(gh, gw, s_freqs)
    h = patch_embed(x.reshape(B * T, C, H, W), sd["x_embedder.proj.weight"], sd["x_embedder.proj.bias"],
                    cfg.patch_size)
    h = h.reshape(B, T, gh, gw, -1)
    c = dit_cond(sd, cfg, t, external_cond)
    if taps is not None:
        taps["c"] = c
        taps["embed"] = h
    for i in range(cfg.depth):
        h = dit_block(sd, cfg, i, h, c, s_angles, t_freqs)
        if taps is not None:
            taps[f"block{i}"] = h
    m = F.linear(F.silu(c), sd["final_layer.adaLN_modulation.1.weight"], sd["final_layer.adaLN_modulation.1.bias"])
    shift, scale = m.chunk(2, dim=-1)
    h = modulate(_ln(h), shift, scale)
    h = F.linear(h, sd["final_layer.linear.weight"], sd["final_layer.linear.bias"])
    out = dit_unpatchify(h.reshape(B * T, gh, gw, -1), cfg)
    return out.reshape(B, T, C, H, W)


# ----------------------------------------------------------------------------------------------
# ViT-VAE (model/vae.py)
# ----------------------------------------------------------------------------------------------
def vae_rope_angles(cfg: VAEConfig, heads: int, dim: int) -> Tensor:
    """model/vae.py:71-76: RotaryEmbedding(dim=head_dim//4, pixel, max_freq=H*W).get_axial_freqs(H,W)
    -> (H, W, head_dim/2): only the first half of each head is rotated."""
    head_dim = dim // heads
    freqs = rope_freqs_pixel(head_dim // 4, cfg.seq_h * cfg.seq_w)
    return rope_angles_axial(cfg.seq_h, cfg.seq_w, freqs)


def vae_attention(sd, pre: str, x: Tensor, heads: int, angles: Tensor, gh: int, gw: int) -> Tensor:
    """model/vae.py:78-112: qkv WITH bias, layout (B,N,3,h,d); partial RoPE; SDPA non-causal; proj."""
    B, N, C = x.shape
    d = C // heads
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"]).reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = apply_rope(angles, q.reshape(B, heads, gh, gw, d)).reshape(B, heads, N, d)
    k = apply_rope(angles, k.reshape(B, heads, gh, gw, d)).reshape(B, heads, N, d)
    o = F.scaled_dot_product_attention(q, k, v, is_causal=False)
    o = o.transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def vae_block(sd, pre: str, x: Tensor, heads: int, angles: Tensor, gh: int, gw: int) -> Tensor:
    """model/vae.py:154-157 AttentionBlock.forward (pre-LN affine, erf-GELU MLP)."""
    x = x + vae_attention(sd, pre + "attn.", _ln(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"]),
                          heads, angles, gh, gw)
    x = x + mlp(sd, pre + "mlp.", _ln(x, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"]), "none")
    return x


def vae_encode_moments(sd, cfg: VAEConfig, x: Tensor) -> Tensor:
    """model/vae.py:306-322 up to `moments` (N, seq_len, 2*latent). x (N,3,H,W) in [-1,1]."""
    h = patch_embed(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], cfg.patch_size)
    h = h.reshape(h.shape[0], -1, h.shape[-1])
    ang = vae_rope_angles(cfg, cfg.enc_heads, cfg.enc_dim)
    for i in range(cfg.enc_depth):
        h = vae_block(sd, f"encoder.{i}.", h, cfg.enc_heads, ang, cfg.seq_h, cfg.seq_w)
    h = _ln(h, sd["enc_norm.weight"], sd["enc_norm.bias"])
    return F.linear(h, sd["quant_conv.weight"], sd["quant_conv.bias"])


def vae_encode_mean(sd, cfg: VAEConfig, x: Tensor) -> Tensor:
    """`vae.encode(x).mean` (model/vae.py:19-22): first latent_dim channels of the moments."""
    return vae_encode_moments(sd, cfg, x)[..., : cfg.latent_dim]


def vae_unpatchify(x: Tensor, cfg: VAEConfig) -> Tensor:
    """model/vae.py:279-304: patch vector ordered (c, p_row, p_col)."""
    N, p = x.shape[0], cfg.patch_size
    x = x.reshape(N, cfg.seq_h, cfg.seq_w, cfg.patch_dim).permute(0, 3, 1, 2)
    x = x.reshape(N, 3, p, p, cfg.seq_h, cfg.seq_w).permute(0, 1, 4, 2, 5, 3)
    return x.reshape(N, 3, cfg.input_height, cfg.input_width)


def vae_decode(sd, cfg: VAEConfig, z: Tensor) -> Tensor:
    """model/vae.py:324-338. z (N, seq_len, latent) -> (N,3,H,W)."""
    h = F.linear(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    ang = vae_rope_angles(cfg, cfg.dec_heads, cfg.dec_dim)
    for i in range(cfg.dec_depth):
        h = vae_block(sd, f"decoder.{i}.", h, cfg.dec_heads, ang, cfg.seq_h, cfg.seq_w)
    h = _ln(h, sd["dec_norm.weight"], sd["dec_norm.bias"])
    h = F.linear(h, sd["predictor.weight"], sd["predictor.bias"])
    return vae_unpatchify(h, cfg)


# ----------------------------------------------------------------------------------------------
# sampler (train_dit.py:30-125) and drivers (generate.py:50-66,186-244; train_dit.py:554-682)
# ----------------------------------------------------------------------------------------------
def denoise_step(dit_fn: Callable, x_noisy: Tensor, actions: Optional[Tensor], noise_idx: int,
                 stabilization_level: int, noise_range: Tensor, alphas_cumprod: Tensor, start_frame: int = 0):
    """train_dit.py:30-125. alphas_cumprod has shape (1000,1,1,1). Returns (x_pred, v_pred) for the
    window x_noisy[:, start_frame:]."""
    B, n = x_noisy.shape[:2]
    t_ctx = torch.full((B, n - 1), int(stabilization_level), dtype=torch.long)
    t = torch.full((B, 1), int(noise_range[noise_idx]), dtype=torch.long)          # long() truncates
    t_next = torch.full((B, 1), int(noise_range[max(0, noise_idx - 1)]), dtype=torch.long)
    t = torch.cat([t_ctx, t], dim=1)[:, start_frame:]
    t_next = torch.cat([t_ctx, t_next], dim=1)[:, start_frame:]
    x_curr = x_noisy.clone()[:, start_frame:]
    if actions is not None:
        actions = actions[:, start_frame: start_frame + x_curr.shape[1]]
    v_pred = dit_fn(x_curr, t, actions)
    alpha_t = alphas_cumprod[t]
    x_start = alpha_t.sqrt() * x_curr - (1 - alpha_t).sqrt() * v_pred
    x_noise = ((1 / alpha_t).sqrt() * x_curr - x_start) / (1 / alpha_t - 1).sqrt()
    alpha_next = alphas_cumprod[t_next].clone()
    alpha_next[:, :-1] = torch.ones_like(alpha_next[:, :-1])
    if noise_idx <= 0:
        return x_start, v_pred
    x_pred = alpha_next.sqrt() * x_start + (1 - alpha_next).sqrt() * x_noise
    return x_pred, v_pred


def vae_encode_frames(vae_sd, vcfg: VAEConfig, frames: Tensor) -> Tensor:
    """generate.py:50-66 / train_dit.py:329-351: frames (B,t,3,H,W) in [0,1] -> latents (B,t,C,h,w)."""
    B, t = frames.shape[:2]
    m = vae_encode_mean(vae_sd, vcfg, frames.reshape(B * t, *frames.shape[2:]) * 2 - 1) * SCALING_FACTOR
    return m.reshape(B, t, vcfg.seq_h, vcfg.seq_w, vcfg.latent_dim).permute(0, 1, 4, 2, 3).contiguous()


def vae_decode_latents(vae_sd, vcfg: VAEConfig, x: Tensor) -> Tensor:
    """generate.py:238-244: latents (B,t,C,h,w) -> uint8 frames (B,t,H,W,3)."""
    B, t, C, h, w = x.shape
    z = x.permute(0, 1, 3, 4, 2).reshape(B * t, h * w, C)
    img = (vae_decode(vae_sd, vcfg, z / SCALING_FACTOR) + 1) / 2
    img = img.reshape(B, t, 3, vcfg.input_height, vcfg.input_width).permute(0, 1, 3, 4, 2)
    return torch.clamp(img * 255, 0, 255).byte()


def generate_latents(dit_fn: Callable, x_prompt: Tensor, total_frames: int, noise_steps: int,
                     noise_chunks: Tensor, actions: Optional[Tensor] = None, max_frames: int = 5,
                     stabilization_level: int = 15, noise_abs_max: float = 20.0, clamp_min: float = 1e-4):
    """generate.py:186-220 sampling loop, batch-generic, with the per-frame initial noise injected
    (`noise_chunks` (B, total-n_prompt, C, h, w)) instead of drawn from an unseeded RNG."""
    x = x_prompt.clone()
    n_prompt = x.shape[1]
    noise_range = noise_range_generate(noise_steps)
    ac = alphas_cumprod_table(clamp_min)[:, None, None, None]
    for i in range(n_prompt, total_frames):
        chunk = torch.clamp(noise_chunks[:, i - n_prompt: i - n_prompt + 1], -noise_abs_max, noise_abs_max)
        x = torch.cat([x, chunk], dim=1)
        start = max(0, i + 1 - max_frames)
        for noise_idx in reversed(range(0, noise_steps + 1)):
            x_pred, _ = denoise_step(dit_fn, x, actions, noise_idx, stabilization_level, noise_range, ac, start)
            x[:, -1:] = x_pred[:, -1:]
    return x


def train_forward_loss(dit_fn: Callable, latents: Tensor, actions: Optional[Tensor], target_noise_idx: Tensor,
                       ctx_noise_idx: Tensor, ctx_noise: Tensor, noise: Tensor, noise_steps: int = 50,
                       n_prompt_frames: int = 4, max_frames: int = 5, noise_abs_max: float = 20.0,
                       clamp_min: float = 1e-6):
    """train_dit.py:554-682 `_shared_step` forward + loss for the shipped 5-frame clips (one loop
    iteration, i = n_prompt_frames), with the random draws injected:
    target_noise_idx, ctx_noise_idx (B,) ints; ctx_noise (B,W-1,C,h,w); noise (B,1,C,h,w).
    Returns (loss, v_pred, v_target, x_noisy, t)."""
    B, total = latents.shape[:2]
    assert total == n_prompt_frames + 1
    nr = noise_range_train(noise_steps)
    ac = alphas_cumprod_table(clamp_min)[:, None, None, None]
    i = n_prompt_frames
    ctx_noise_idx = torch.minimum(ctx_noise_idx, target_noise_idx)
    start = max(0, i + 1 - max_frames)
    t = torch.zeros((B, i + 1), dtype=torch.long)
    t[:, :-1] = nr[ctx_noise_idx].unsqueeze(1)
    t[:, -1] = nr[target_noise_idx]
    x_curr = latents[:, start: i + 1]
    t = t[:, start:]
    a = actions[:, start: i + 1] if actions is not None else None
    ctx_noise = ctx_noise.clamp(-noise_abs_max, noise_abs_max)
    noise = noise.clamp(-noise_abs_max, noise_abs_max)
    x_noisy = x_curr.clone()
    al = ac[t[:, :-1]]
    x_noisy[:, :-1] = x_noisy[:, :-1] * al.sqrt() + (1 - al).sqrt() * ctx_noise
    al = ac[t[:, -1:]]
    x_noisy[:, -1:] = x_noisy[:, -1:] * al.sqrt() + (1 - al).sqrt() * noise
    v_target = al.sqrt() * noise - (1 - al).sqrt() * x_curr[:, -1:]
    v_pred = dit_fn(x_noisy, t, a)
    loss = F.mse_loss(v_pred[:, -1:], v_target)
    return loss, v_pred, v_target, x_noisy, t


def train_shared_step(dit_fn: Callable, latents: Tensor, actions: Optional[Tensor], target_noise_idx: Tensor, ctx_noise_idx: Tensor,
                      ctx_noises, noises, noise_steps: int = 50, n_prompt_frames: int = 4, max_frames: int = 5,
                      noise_abs_max: float = 20.0, clamp_min: float = 1e-6):
    """train_dit.py:590-682 — the whole frame loop of `_shared_step` for clips with any number of target frames, random draws injected:
    target_noise_idx / ctx_noise_idx (n, B) ints, ctx_noises[k] (B, W_k - 1, C, h, w), noises[k] (B, 1, C, h, w), n = total_frames -
    n_prompt_frames, W_k = min(n_prompt_frames + k + 1, max_frames).  Returns (mean loss over the target frames, [per-frame loss],
    [v_pred], [v_target])."""
    B, total = latents.shape[:2]
    n_iter = total - n_prompt_frames
    nr = noise_range_train(noise_steps)
    ac = alphas_cumprod_table(clamp_min)[:, None, None, None]
    ctx_noise_idx = torch.minimum(ctx_noise_idx, target_noise_idx)                      # :587
    losses, vps, vts = [], [], []
    for idx, i in enumerate(range(n_prompt_frames, total)):                              # :590
        start = max(0, i + 1 - max_frames)                                               # :598
        t = torch.zeros((B, i + 1), dtype=torch.long)
        t[:, :-1] = nr[ctx_noise_idx[idx]].unsqueeze(1)                                  # :610-611
        t[:, -1] = nr[target_noise_idx[idx]]
        x_curr, t = latents[:, start: i + 1], t[:, start:]                               # :614-615
        a = actions[:, start: i + 1] if actions is not None else None
        cn = ctx_noises[idx].clamp(-noise_abs_max, noise_abs_max)
        nz = noises[idx].clamp(-noise_abs_max, noise_abs_max)
        x_noisy = x_curr.clone()
        al = ac[t[:, :-1]]
        x_noisy[:, :-1] = x_noisy[:, :-1] * al.sqrt() + (1 - al).sqrt() * cn             # :631-634
        al = ac[t[:, -1:]]
        x_noisy[:, -1:] = x_noisy[:, -1:] * al.sqrt() + (1 - al).sqrt() * nz             # :639-643
        v_target = al.sqrt() * nz - (1 - al).sqrt() * x_curr[:, -1:]                     # :644-646
        v_pred = dit_fn(x_noisy, t, a)                                                   # :649
        losses.append(F.mse_loss(v_pred[:, -1:], v_target))                              # :650
        vps.append(v_pred)
        vts.append(v_target)
    return sum(losses) / n_iter, losses, vps, vts                                        # :676, :682


def trainer_predict_latents(dit_fn: Callable, vae_sd, vcfg: VAEConfig, frames: Tensor, actions: Optional[Tensor],
                            new_frame_noise: Tensor, num_frames: int, n_prompt_frames: int = 4, ddim_noise_steps: int = 50,
                            ddim_noise_steps_inference: int = 50, max_frames: int = 5, noise_abs_max: float = 20.0) -> Tensor:
    """train_dit.py:370-466 `predict` up to the latents (the decode tail is `vae_decode_latents`): trainer constants — schedule
    clamp_min 1e-6 (:292), stabilization_level = noise_range[1] of the training range (:327), long() inference range (:315)."""
    frames = frames[:1, :n_prompt_frames]
    if actions is not None:
        actions = actions[:1]
        if actions.shape[1] < num_frames:
            pad = torch.zeros((1, num_frames - actions.shape[1], actions.shape[2]))
            pad[:, :, 3] = 1
            actions = torch.cat([actions, pad], dim=1)
    x = vae_encode_frames(vae_sd, vcfg, frames)
    lvl = int(noise_range_train(ddim_noise_steps)[1])
    nri = noise_range_train(ddim_noise_steps_inference)
    ac = alphas_cumprod_table(1e-6)[:, None, None, None]
    for i in range(n_prompt_frames, num_frames):
        chunk = torch.clamp(new_frame_noise[:, i - n_prompt_frames: i - n_prompt_frames + 1], -noise_abs_max, noise_abs_max)
        x = torch.cat([x, chunk], dim=1)
        start = max(0, i + 1 - max_frames)
        for noise_idx in reversed(range(0, ddim_noise_steps_inference + 1)):
            x_pred, _ = denoise_step(dit_fn, x, actions, noise_idx, lvl, nri, ac, start)
            x[:, -1:] = x_pred[:, -1:]
    return x


def trainer_predict_noise(dit_fn: Callable, vae_sd, vcfg: VAEConfig, frames: Tensor, actions: Optional[Tensor], ctx_noise: Tensor,
                          new_frame_noise: Tensor, ddim_noise_steps: int = 50, ddim_noise_steps_inference: int = 50,
                          max_frames: int = 5, noise_abs_max: float = 20.0):
    """train_dit.py:468-552 `predict_noise`: returns (latents, x_noisy before denoising, x_noisy after)."""
    latents = vae_encode_frames(vae_sd, vcfg, frames[:1])
    n = latents.shape[1]
    lvl = int(noise_range_train(ddim_noise_steps)[1])
    nri = noise_range_train(ddim_noise_steps_inference)
    ac = alphas_cumprod_table(1e-6)[:, None, None, None]
    x_noisy = latents.clone()
    cn = torch.clamp(ctx_noise, -noise_abs_max, noise_abs_max)
    t_ctx = torch.full((1, n - 1), lvl - 1, dtype=torch.long)                       # train_dit.py:495-501
    a = ac[t_ctx]
    x_noisy[:, :-1] = a.sqrt() * x_noisy[:, :-1] + (1 - a).sqrt() * cn
    x_noisy[:, -1:] = torch.clamp(new_frame_noise, -noise_abs_max, noise_abs_max)
    if actions is not None:
        actions = actions[:1]
    start = max(0, n - max_frames)
    x_old = x_noisy.clone()
    for noise_idx in reversed(range(0, ddim_noise_steps_inference + 1)):
        x_pred, _ = denoise_step(dit_fn, x_noisy, actions, noise_idx, lvl, nri, ac, start)
        x_noisy[:, -1:] = x_pred[:, -1:]
    return latents, x_old, x_noisy


# ----------------------------------------------------------------------------------------------
# synthetic inputs (dummy_dataset.py:15-36, web_dataset.py:22-38)
# ----------------------------------------------------------------------------------------------
def resize_frames(x: Tensor, size=(360, 640)) -> Tensor:
    """torchvision.transforms.Resize(size) on a float tensor (..., H, W) (web_dataset.py:107, hf_dataset.py:32, generate.py:151):
    torchvision.transforms.functional.resize -> torch.nn.functional.interpolate(mode="bilinear", align_corners=False,
    antialias=True).  torchvision is absent from the image (parity unpinned for this row): this calls the same ATen kernel
    torchvision would; for the dataset's up-scaling (270x480 -> 360x640) antialiasing is a no-op."""
    lead = x.shape[:-3]
    y = F.interpolate(x.reshape(-1, *x.shape[-3:]).float(), size=tuple(size), mode="bilinear", align_corners=False, antialias=True)
    return y.reshape(*lead, *y.shape[-3:])


def strip_to_clip(strip_u8_hwc: Tensor, n_frames=5, size=(360, 640)) -> Tensor:
    """web_dataset.py:41-57,105-107 / hf_dataset.py:30-33: Compose([ToTensor(), SplitImages(), Resize(size)]) on a decoded strip
    image (H, n*W, 3) uint8: -> (n, 3, size[0], size[1]) float in [0, 1]."""
    img = strip_u8_hwc.permute(2, 0, 1).float() / 255.0                    # ToTensor
    c, h, wt = img.shape
    w = wt // n_frames
    clip = img.reshape(c, h, n_frames, w).permute(2, 0, 1, 3)              # SplitImages: "c h (n w) -> n c h w"
    return resize_frames(clip, size)


def dummy_clip(height=360, width=640, n=5) -> Tensor:
    """dummy_dataset.py:15-25: n constant-colour frames blue->red, (n,3,H,W) in [0,1]."""
    blue, red = torch.tensor([0.0, 0.0, 1.0]), torch.tensor([1.0, 0.0, 0.0])
    frames = [((1 - s) * blue + s * red).view(3, 1, 1).expand(3, height, width) for s in torch.linspace(0, 1, n)]
    return torch.stack(frames).contiguous()


def actions_to_one_hot(actions) -> Tensor:
    """web_dataset.py:22-38: 25-way one-hot, -1 -> zero row."""
    a = torch.as_tensor(actions)
    out = torch.zeros(len(a), 25)
    m = a >= 0
    out[torch.arange(len(a))[m], a[m]] = 1
    return out


# ------------------------------------------------------------------------------------------------------------------------
# Training step oracle: the reference obtains gradients from torch autograd (train_dit.py:680
# `accelerator.backward(scaled_loss)`) and updates with torch.optim.AdamW (:232-238) after clip_grad_norm_ (:965-967).  Autograd
# through this file's functional forward IS that computation on the CPU in fp32; nothing is re-derived by hand here.
# ------------------------------------------------------------------------------------------------------------------------
def dit_loss_and_grads(sd: Dict[str, Tensor], cfg: DiTConfig, x_noisy: Tensor, t: Tensor, actions: Optional[Tensor], v_target: Tensor):
    """loss = mse(v_pred[:, -1:], v_target) (train_dit.py:649-650) and d loss / d every floating-point parameter but the rotary freqs."""
    leaves = {k: v.detach().clone().requires_grad_(not k.endswith("freqs")) for k, v in sd.items()}
    with torch.enable_grad():
        v_pred = dit_forward(leaves, cfg, x_noisy, t, actions)
        loss = torch.nn.functional.mse_loss(v_pred[:, -1:], v_target)
        loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items() if not k.endswith("freqs")}
    return loss.detach(), v_pred.detach(), grads


def adamw_reference(params: Dict[str, Tensor], grads: Dict[str, Tensor], lr: float, weight_decay: float, max_grad_norm: float, steps: int = 1):
    """clip_grad_norm_ + torch.optim.AdamW(betas=(0.9, 0.999), eps=1e-7) applied `steps` times with the same gradients."""
    ps = [torch.nn.Parameter(v.detach().clone()) for v in params.values()]
    opt = torch.optim.AdamW(ps, lr=lr, weight_decay=weight_decay, betas=(0.9, 0.999), eps=1e-7)
    norm = None
    for _ in range(steps):
        for p, g in zip(ps, grads.values()):
            p.grad = g.detach().clone()
        norm = torch.nn.utils.clip_grad_norm_(ps, max_grad_norm) if max_grad_norm > 0 else None
        opt.step()
    return {k: p.detach() for k, p in zip(params.keys(), ps)}, norm
