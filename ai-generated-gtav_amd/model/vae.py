"""Drop-in mirror of the reference `model/vae.py` public surface (model/vae.py:19-45,160-384):
`AutoencoderKL`, `DiagonalGaussianDistribution`, `ViT_L_20_Shallow_Encoder`, `VAE_models` — same
constructor signature, `encode(x).mean` / `decode(z)` call shapes and state-dict names; compute runs
in libgtav_amd.so (`gtav_vae_*`)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from .. import lib as _lib
from .. import weights as _w
from .dit import _HipModule, _rope_tables_axial


class DiagonalGaussianDistribution:
    """model/vae.py:19-45 for dim=2 parameters (N, seq_len, 2*latent). `logvar` arrives already clamped to
    [-30, 20] by the encode kernel; std/var/sample are lazy (not on the hot path)."""

    def __init__(self, parameters: torch.Tensor, deterministic: bool = False, dim: int = 2):
        assert dim == 2
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=dim)
        self.deterministic = deterministic

    @property
    def std(self):
        return torch.zeros_like(self.mean) if self.deterministic else torch.exp(0.5 * self.logvar)

    @property
    def var(self):
        return torch.zeros_like(self.mean) if self.deterministic else torch.exp(self.logvar)

    def sample(self):
        return self.mean + self.std * torch.randn(self.mean.shape).to(device=self.parameters.device)

    def mode(self):
        return self.mean


class AutoencoderKL(_HipModule):
    """model/vae.py:160-361."""

    _prefix = "gtav_vae"

    def __init__(self, latent_dim, input_height=270, input_width=480, patch_size=24, enc_dim=768, enc_depth=6, enc_heads=12,
                 dec_dim=768, dec_depth=6, dec_heads=12, mlp_ratio=4.0, norm_layer=None, use_variational=True,
                 *, max_frames_per_call=8, grow_frames_per_call=False, init_weights=True, **kwargs):
        """`max_frames_per_call` (keyword-only, not in the reference) sizes the handle's HBM workspace: encode / decode of more frames run in chunks of that
        many (frames never interact in the VAE, model/vae.py:306-338).  The workspace costs ~23 MB per frame at ViT-L/20 on 360 x 640 (1.8 GB at 80 frames per
        call) and the chunk size selects the kernels (from 80 frames per call on: 256 x 256 in-place residual tiles, one flash-attention grid), so results
        move by ~3e-4 relative L2 between chunk sizes — which is why the cap is the caller's choice.  `grow_frames_per_call=True` lets a larger call re-create
        the handle with room for it (up to 128 frames) instead of chunking."""
        super().__init__()
        self._grow = bool(grow_frames_per_call)
        self.input_height, self.input_width, self.patch_size = input_height, input_width, patch_size
        self.seq_h, self.seq_w = input_height // patch_size, input_width // patch_size
        self.seq_len = self.seq_h * self.seq_w
        self.patch_dim = 3 * patch_size ** 2
        self.latent_dim, self.enc_dim, self.dec_dim = latent_dim, enc_dim, dec_dim
        self.use_variational = use_variational
        self._cfg_kwargs = dict(latent_dim=latent_dim, input_height=input_height, input_width=input_width,
                                patch_size=patch_size, enc_dim=enc_dim, enc_depth=enc_depth, enc_heads=enc_heads,
                                dec_dim=dec_dim, dec_depth=dec_depth, dec_heads=dec_heads, mlp_ratio=mlp_ratio,
                                use_variational=int(bool(use_variational)))
        self._enc_heads, self._dec_heads = enc_heads, dec_heads
        self._capacity = max_frames_per_call
        if init_weights:
            self.initialize_weights()

    def _shapes(self):
        return _w.vae_param_shapes(**self._cfg_kwargs)

    def initialize_weights(self):
        """Distribution-identical restatement of model/vae.py:239-256 (xavier-uniform linears, zero biases,
        LayerNorm weight 1 / bias 0)."""
        for k, shp in self._shapes().items():
            if k == "patch_embed.proj.bias":
                # the only bias `_init_weights` does not reach (a Conv2d): it keeps torch's default U(+-1/sqrt(fan_in))
                b = 1.0 / (3 * self.patch_size ** 2) ** 0.5
                v = (torch.rand(shp) * 2 - 1) * b
            elif k.endswith(".bias"):
                v = torch.zeros(shp)
            elif "norm" in k and k.endswith(".weight"):
                v = torch.ones(shp)
            else:
                fan_out = shp[0]
                fan_in = 1
                for d in shp[1:]:
                    fan_in *= d
                bound = (6.0 / (fan_in + fan_out)) ** 0.5
                v = (torch.rand(shp) * 2 - 1) * bound
            self._sd[k] = v
        self._dirty = True

    def _ensure(self, n: int):
        want = min(max(n, 1), 128)
        if want > self._capacity and (self._grow or not self._handle and self._capacity < 1):
            self._capacity = want
            self._free()
        if not self._handle:
            cfg = _lib.VaeConfig(max_frames_per_call=self._capacity, **self._cfg_kwargs)
            with torch.cuda.device(self.device):
                _lib.check(_lib.load().gtav_vae_create(C.byref(cfg), C.byref(self._handle)))
                if getattr(self, "_bf16", False):
                    _lib.check(_lib.load().gtav_vae_set_operand_dtype(self._handle, 1))
            self._dirty = True
        if self._dirty:
            # model/vae.py:71-76: RotaryEmbedding(dim=head_dim // 4, pixel, max_freq=H*W).get_axial_freqs(H, W)
            extra = {}
            for tag, dim, heads in (("enc", self.enc_dim, self._enc_heads), ("dec", self.dec_dim, self._dec_heads)):
                fr = _w.rope_freqs_pixel((dim // heads) // 4, self.seq_h * self.seq_w)
                c, s = _rope_tables_axial(fr, self.seq_h, self.seq_w)
                extra[f"tables.rope_{tag}_cos"], extra[f"tables.rope_{tag}_sin"] = c, s
            self._upload(extra)

    # ------------------------------------------------------------------------------------------
    def encode_moments(self, x: torch.Tensor, in_scale: float = 1.0, in_shift: float = 0.0) -> torch.Tensor:
        """quant_conv output (N, seq_len, 2*latent), logvar clamped; the network sees in_scale*x + in_shift."""
        N = x.shape[0]
        assert x.shape[1:] == (3, self.input_height, self.input_width), (
            f"Input image size ({x.shape[-2]}*{x.shape[-1]}) doesn't match model ({self.input_height}*{self.input_width}).")
        self._ensure(N)
        dev = self.device
        xd = x.to(dev, torch.float32).contiguous()
        mom_ch = (2 if self.use_variational else 1) * self.latent_dim
        out = torch.empty((N, self.seq_len, mom_ch), device=dev, dtype=torch.float32)
        L = _lib.load()
        with torch.cuda.device(dev):
            for i in range(0, N, self._capacity):
                n = min(self._capacity, N - i)
                _lib.check(L.gtav_vae_encode(self._handle, xd[i:i + n].data_ptr(), in_scale, in_shift, out[i:i + n].data_ptr(),
                                             n, _lib.current_stream()))
        return out

    def encode(self, x: torch.Tensor) -> DiagonalGaussianDistribution:
        """model/vae.py:306-322."""
        moments = self.encode_moments(x)
        if not self.use_variational:
            moments = torch.cat((moments, torch.zeros_like(moments)), 2)
        return DiagonalGaussianDistribution(moments, deterministic=(not self.use_variational), dim=2)

    def decode(self, z: torch.Tensor, z_scale: float = 1.0, out_scale: float = 1.0, out_shift: float = 0.0) -> torch.Tensor:
        """model/vae.py:324-338: z (N, seq_len, latent) -> (N, 3, H, W); optional fused affine in/out."""
        N = z.shape[0]
        assert z.shape[1:] == (self.seq_len, self.latent_dim)
        self._ensure(N)
        dev = self.device
        zd = z.to(dev, torch.float32).contiguous()
        out = torch.empty((N, 3, self.input_height, self.input_width), device=dev, dtype=torch.float32)
        L = _lib.load()
        with torch.cuda.device(dev):
            for i in range(0, N, self._capacity):
                n = min(self._capacity, N - i)
                _lib.check(L.gtav_vae_decode(self._handle, zd[i:i + n].data_ptr(), z_scale, out[i:i + n].data_ptr(), out_scale,
                                             out_shift, n, _lib.current_stream()))
        return out

    def set_operand_dtype(self, dtype):
        """torch.float16 (default) or torch.bfloat16 (the reference's autocast type: fp32 range, lower precision) for every 2-byte GEMM / attention operand of
        the VAE (include/gtav_amd.h "operand type"); the weights are converted again from the fp32 host copies at the next call."""
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError(f"set_operand_dtype: {dtype} (torch.float16 or torch.bfloat16)")
        bf = dtype == torch.bfloat16
        if bf == getattr(self, "_bf16", False):
            return
        self._bf16 = bf
        if self._handle:
            _lib.check(_lib.load().gtav_vae_set_operand_dtype(self._handle, 1 if bf else 0))
            self._dirty = True

    def check(self):
        """Raises if an input held NaN/inf or an fp16 activation saturated since the last call (gtav_vae_check; synchronises)."""
        if self._handle:
            with torch.cuda.device(self.device):
                _lib.check(_lib.load().gtav_vae_check(self._handle, _lib.current_stream()))

    PROFILE_CLASSES = ("ln_affine", "gemm_qkv", "attn_spatial", "attn_temporal", "gemm_proj", "gemm_fc1", "gemm_fc2", "other", "empty_event_pair")

    def profile(self, enable: bool):
        """In-situ per-kernel-class HIP-event timing of encode / decode (measurement passes only; include/gtav_amd.h gtav_vae_profile)."""
        self._ensure(1)
        _lib.check(_lib.load().gtav_vae_profile(self._handle, int(bool(enable))))

    def profile_read(self):
        import ctypes as C
        ms = (C.c_double * 9)()
        n = (C.c_int64 * 9)()
        _lib.check(_lib.load().gtav_vae_profile_read(self._handle, ms, n))
        return {k: (ms[i], n[i]) for i, k in enumerate(self.PROFILE_CLASSES)}

    def autoencode(self, input, sample_posterior=True):
        """model/vae.py:340-347."""
        posterior = self.encode(input)
        z = posterior.sample() if (self.use_variational and sample_posterior) else posterior.mode()
        return self.decode(z), posterior, z

    def forward(self, inputs, labels=None, split="train"):
        return self.autoencode(inputs)

    __call__ = forward

    # layout-only helpers of the reference (model/vae.py:258-304), pure reshapes
    def patchify(self, x):
        b = x.shape[0]
        p = self.patch_size
        x = x.reshape(b, 3, self.seq_h, p, self.seq_w, p).permute(0, 1, 3, 5, 2, 4)
        return x.reshape(b, self.patch_dim, self.seq_h, self.seq_w).permute(0, 2, 3, 1).reshape(b, self.seq_len, self.patch_dim)

    def unpatchify(self, x):
        b = x.shape[0]
        p = self.patch_size
        x = x.reshape(b, self.seq_h, self.seq_w, self.patch_dim).permute(0, 3, 1, 2)
        x = x.reshape(b, 3, p, p, self.seq_h, self.seq_w).permute(0, 1, 4, 2, 5, 3)
        return x.reshape(b, 3, self.input_height, self.input_width)


def ViT_L_20_Shallow_Encoder(**kwargs):
    """model/vae.py:363-380."""
    latent_dim = kwargs.pop("latent_dim", 16)
    return AutoencoderKL(latent_dim=latent_dim, patch_size=20, enc_dim=1024, enc_depth=6, enc_heads=16, dec_dim=1024,
                         dec_depth=12, dec_heads=16, input_height=360, input_width=640, **kwargs)


VAE_models = {"vit-l-20-shallow-encoder": ViT_L_20_Shallow_Encoder}
