"""Drop-in mirror of the reference `model/dit.py` public surface (model/dit.py:228-392): `DiT`,
`DiT_S_2`, `DiT_models` — same constructor signature, `forward(x, t, external_cond)` call shape,
`max_frames` / `patch_size` attributes and state-dict names — with every FLOP executed by the HIP
kernels of libgtav_amd.so through the C-ABI (`gtav_dit_*`, include/gtav_amd.h).  torch is used for
storage (device tensors, the fp32 master copy of the weights) and host-side constant tables only.
"""
from __future__ import annotations

import ctypes as C
import math
from collections import OrderedDict
from typing import Dict, Iterator, Optional

import torch

from .. import lib as _lib
from .. import weights as _w


def _rope_tables_axial(freqs: torch.Tensor, gh: int, gw: int):
    """cos/sin tables (gh*gw, 64) of `RotaryEmbedding.get_axial_freqs(gh, gw)` for pixel freqs
    (model/rotary_embedding_torch.py:290-345), identity beyond the rotated dims."""
    def ang(n):
        a = torch.linspace(-1, 1, steps=n)[:, None] * freqs[None, :]
        return a.repeat_interleave(2, dim=-1)
    ah, aw = ang(gh), ang(gw)
    a = torch.cat([ah[:, None, :].expand(gh, gw, -1), aw[None, :, :].expand(gh, gw, -1)], dim=-1).reshape(gh * gw, -1)
    cos, sin = torch.ones(gh * gw, 64), torch.zeros(gh * gw, 64)
    cos[:, : a.shape[1]] = a.cos()
    sin[:, : a.shape[1]] = a.sin()
    return cos.contiguous(), sin.contiguous()


def _rope_tables_temporal(freqs: torch.Tensor, T: int):
    """`rotate_queries_or_keys` angles for positions 0..T-1 (rotary_embedding_torch.py:186-209)."""
    a = (torch.arange(T, dtype=torch.float32)[:, None] * freqs[None, :]).repeat_interleave(2, dim=-1)
    return a.cos().contiguous(), a.sin().contiguous()


def _timestep_table() -> torch.Tensor:
    """TimestepEmbedder.timestep_embedding for every t in [0, 999] (model/dit.py:96-118): (1000, 256)."""
    half = 128
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
    args = torch.arange(1000, dtype=torch.float32)[:, None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1).contiguous()


class _HipModule:
    """Shared plumbing of DiT / AutoencoderKL: fp32 master weights on the host, a C-ABI handle on the GPU."""

    _prefix = ""  # gtav_dit / gtav_vae

    def __init__(self):
        self._sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self._handle = C.c_void_p(None)
        self._dirty = True
        self.training = False
        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cuda")

    # --- nn.Module-like surface the reference callers use (SURVEY.md §8(b) "Attributes read") ---
    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def to(self, *args, **kwargs):
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, (str, torch.device)) and torch.device(a).type == "cuda":
                self.device = torch.device(a)
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", device if device is not None else torch.cuda.current_device()))

    def parameters(self) -> Iterator[torch.Tensor]:
        return iter(self._sd.values())

    def named_parameters(self):
        return iter(self._sd.items())

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        sd = OrderedDict(self._sd)
        sd.update(self._extra_state())
        return sd

    def _extra_state(self) -> Dict[str, torch.Tensor]:
        return {}

    def _shapes(self):
        raise NotImplementedError

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        params, sf, tf = _w.split_freq_keys(dict(sd))
        shapes = self._shapes()
        missing = [k for k in shapes if k not in params]
        unexpected = [k for k in params if k not in shapes]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing keys {missing[:5]}..., unexpected keys {unexpected[:5]}...")
        for k, shp in shapes.items():
            if k in params:
                v = params[k].detach().to("cpu", torch.float32).contiguous()
                if tuple(v.shape) != tuple(shp):
                    raise RuntimeError(f"load_state_dict: {k} has shape {tuple(v.shape)}, expected {tuple(shp)}")
                self._sd[k] = v
        self._load_freqs(sf, tf)
        self._dirty = True
        return missing, unexpected

    def _load_freqs(self, sf, tf):
        pass

    def _free(self):
        if self._handle:
            getattr(_lib.load(), self._prefix + "_destroy")(self._handle)
            self._handle = C.c_void_p(None)

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    def _upload(self, extra: Dict[str, torch.Tensor]):
        L = _lib.load()
        setw = getattr(L, self._prefix + "_set_weight")
        stream = _lib.current_stream()
        with torch.cuda.device(self.device):
            for name, v in list(self._sd.items()) + list(extra.items()):
                d = v.to(self.device, torch.float32, non_blocking=False).contiguous()
                _lib.check(setw(self._handle, name.encode(), d.data_ptr(), d.numel(), stream))
                del d
            _lib.check(getattr(L, self._prefix + "_finalize")(self._handle, stream))
            torch.cuda.synchronize()
        self._dirty = False


class DiT(_HipModule):
    """model/dit.py:228-376.  `max_batch` (keyword-only, not in the reference) pre-sizes the HBM workspace;
    it grows automatically when a larger batch arrives."""

    _prefix = "gtav_dit"

    def __init__(self, input_h=18, input_w=32, patch_size=2, in_channels=16, hidden_size=1024, depth=12, num_heads=16,
                 mlp_ratio=4.0, external_cond_dim=25, max_frames=5, *, max_batch=1, init_weights=True, trainable=False, range_policy="report"):
        super().__init__()
        self._trainable = bool(trainable)    # keyword-only, not in the reference: keeps fp32 masters, gradients and AdamW state on the GPU
        self._grads = None
        self._loss_scale = 65536.0
        self.in_channels = in_channels
        self.out_channels = in_channels
        self.patch_size = patch_size
        self.num_heads = num_heads
        self._max_frames = max_frames
        self.input_h, self.input_w, self.hidden_size, self.depth = input_h, input_w, hidden_size, depth
        self.mlp_ratio, self.external_cond_dim = mlp_ratio, external_cond_dim
        self._cfg_kwargs = dict(input_h=input_h, input_w=input_w, patch_size=patch_size, in_channels=in_channels,
                                hidden_size=hidden_size, depth=depth, num_heads=num_heads, mlp_ratio=mlp_ratio,
                                external_cond_dim=external_cond_dim)
        self._capacity_b, self._capacity_t = max_batch, max(max_frames, 1)
        self._capacity_rows = 0
        hd = hidden_size // num_heads
        self._spatial_freqs = _w.rope_freqs_pixel(hd // 2, 256)   # model/dit.py:259-261
        self._temporal_freqs = _w.rope_freqs_lang(hd)             # model/dit.py:262
        self._schedule = None
        # operand groups (include/gtav_amd.h "operand type"): 2 l / 2 l + 1 = spatial / temporal half of block l, 2 depth = patch embedding, 2 depth + 1 = final layer
        self.n_operand_groups = 2 * depth + 2
        self._bf16_groups = set()
        self.range_policy = range_policy
        if range_policy not in ("report", "auto"):
            raise ValueError("range_policy must be 'report' (check() raises when an fp16 activation saturated) or 'auto' (check() moves the saturated "
                             "layers to bf16 operands and raises GtavRangeSwitch: recompute)")
        if init_weights:
            self.initialize_weights()

    # `max_frames` is read AND assigned by callers (generate.py:139,204)
    @property
    def max_frames(self):
        return self._max_frames

    @max_frames.setter
    def max_frames(self, v):
        self._max_frames = int(v)
        if v > self._capacity_t:
            if self._handle and self._trainable and self._grads is not None:
                raise RuntimeError("a trainable DiT cannot grow its window after training was enabled: construct it with the largest max_frames")
            self._capacity_t = int(v)
            self._free()

    def _shapes(self):
        return _w.dit_param_shapes(**self._cfg_kwargs)

    def _extra_state(self):
        # same de-duplicated aliases the reference's own writer keeps (train_dit.py:758-762)
        return {"spatial_rotary_emb.freqs": self._spatial_freqs, "temporal_rotary_emb.freqs": self._temporal_freqs}

    def _load_freqs(self, sf, tf):
        if sf is not None:
            self._spatial_freqs = sf.detach().float().cpu()
        if tf is not None:
            self._temporal_freqs = tf.detach().float().cpu()

    def initialize_weights(self):
        """Distribution-identical restatement of model/dit.py:295-326 (N(0,.02) linears, zero biases,
        t-MLP std .01, adaLN zeroed, final adaLN std .01, final linear std .001)."""
        for k, shp in self._shapes().items():
            if k.endswith(".bias"):
                v = torch.zeros(shp)
            elif "adaLN_modulation" in k and k.startswith("blocks."):
                v = torch.zeros(shp)
            elif k.startswith("t_embedder.mlp") or k.startswith("final_layer.adaLN_modulation"):
                v = torch.randn(shp) * 0.01
            elif k == "final_layer.linear.weight":
                v = torch.randn(shp) * 0.001
            else:
                v = torch.randn(shp) * 0.02
            self._sd[k] = v
        self._dirty = True

    # ------------------------------------------------------------------------------------------
    def _ensure(self, B: int, T: int, cond_rows: int = 0):
        grow = (cond_rows > max(self._capacity_rows, self._capacity_b * self._capacity_t)) or T > self._capacity_t or B > self._capacity_b
        if grow and self._handle and self._trainable and self._grads is not None:
            # the fp32 masters and the AdamW state live only in the handle: refuse BEFORE anything is destroyed (the handle stays usable)
            raise RuntimeError(f"a trainable DiT cannot grow its workspace after training was enabled (batch {B} > {self._capacity_b}, window {T} > "
                               f"{self._capacity_t} or {cond_rows} conditioning rows > {max(self._capacity_rows, self._capacity_b * self._capacity_t)}): "
                               "size it with max_batch / max_frames / reserve() before the first step")
        if grow:
            self._capacity_rows = max(self._capacity_rows, cond_rows)
            self._capacity_t = max(self._capacity_t, T)
            self._capacity_b = max(self._capacity_b, B)
            self._free()
        if not self._handle:
            L = _lib.load()
            cfg = _lib.DitConfig(max_frames=self._capacity_t, max_batch=self._capacity_b,
                                 max_cond_rows=max(self._capacity_b * self._capacity_t, self._capacity_rows), **self._cfg_kwargs)
            with torch.cuda.device(self.device):
                _lib.check(L.gtav_dit_create(C.byref(cfg), C.byref(self._handle)))
                if getattr(self, "_fused_temporal", False):
                    _lib.check(L.gtav_dit_set_fused_temporal(self._handle, 1))
                if getattr(self, "_fused_spatial", None) is not None:     # None: the library's default (on where the geometry allows)
                    _lib.check(L.gtav_dit_set_fused_spatial(self._handle, int(self._fused_spatial)))
                if getattr(self, "_fold", None) is not None:
                    _lib.check(L.gtav_dit_set_fold(self._handle, *self._fold))
                if getattr(self, "_weight_prefetch", None) is not None:
                    _lib.check(L.gtav_dit_set_weight_prefetch(self._handle, int(self._weight_prefetch)))
                for g in sorted(self._bf16_groups):                      # a re-created handle keeps the operand types chosen before
                    _lib.check(L.gtav_dit_set_operand_dtype(self._handle, g, 1))
                if self._trainable:
                    n = C.c_int64(0)
                    _lib.check(L.gtav_dit_train_param_count(self._handle, C.byref(n)))
                    # one contiguous fp32 gradient arena owned by torch: a single all-reduce covers the whole model (train.py)
                    self._grads = torch.zeros(n.value, device=self.device, dtype=torch.float32)
                    _lib.check(L.gtav_dit_train_enable(self._handle, self._grads.data_ptr(), n.value))
                    _lib.check(L.gtav_dit_set_loss_scale(self._handle, self._loss_scale))
                    if self.grad_divisor != 1.0:
                        _lib.check(L.gtav_dit_set_grad_divisor(self._handle, self.grad_divisor))
            self._dirty = True
        if self._dirty:
            gh, gw = self.input_h // self.patch_size, self.input_w // self.patch_size
            sc, ss = _rope_tables_axial(self._spatial_freqs, gh, gw)
            tc, ts = _rope_tables_temporal(self._temporal_freqs, self._capacity_t)
            extra = {"tables.timestep_sincos": _timestep_table(), "tables.rope_spatial_cos": sc, "tables.rope_spatial_sin": ss,
                     "tables.rope_temporal_cos": tc, "tables.rope_temporal_sin": ts}
            self._upload(extra)
            if self._schedule is not None:
                self.set_schedule(self._schedule)

    def reserve(self, max_batch: int, max_frames: Optional[int] = None, noise_steps: int = 0):
        """Sizes the handle's HBM workspace once for the largest batch / window / per-frame conditioning table a run will use
        (noise_steps + 1 row sets per generated frame), so that no later call has to rebuild the handle (which re-uploads the
        weights and drops the captured hipGraphs)."""
        T = max_frames or self._capacity_t
        self._ensure(max_batch, T, cond_rows=max_batch * (T - 1 + noise_steps + 1) if noise_steps else 0)

    def forward(self, x: torch.Tensor, t: torch.Tensor, external_cond: Optional[torch.Tensor] = None) -> torch.Tensor:
        """model/dit.py:343-376.  x (B,T,C,H,W), t (B,T) integer timesteps, external_cond (B,T,25) or None."""
        B, T, Cc, H, W = x.shape
        assert H == self.input_h and W == self.input_w, (
            f"Input image size ({H}*{W}) doesn't match model ({self.input_h}*{self.input_w}).")  # model/dit.py:67-69
        self._ensure(B, T)
        dev = self.device
        xd = x.to(dev, torch.float32).contiguous()
        td = t.to(dev, torch.int64).contiguous()
        ad = external_cond.to(dev, torch.float32).contiguous() if torch.is_tensor(external_cond) else None
        out = torch.empty_like(xd)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().gtav_dit_forward(self._handle, xd.data_ptr(), td.data_ptr(), _lib.ptr(ad), out.data_ptr(),
                                                    B, T, _lib.current_stream()))
        return out

    __call__ = forward

    # ------------------------------------------------------------------------------------------
    # training step (reference: train_dit.py:649-650, :680, :232-238, :965-970)
    # ------------------------------------------------------------------------------------------
    def forward_train(self, x: torch.Tensor, t: torch.Tensor, external_cond: Optional[torch.Tensor] = None) -> torch.Tensor:
        """DiT.forward keeping the activations `backward_` needs (same result as forward)."""
        assert self._trainable, "construct the model with trainable=True"
        B, T, Cc, H, W = x.shape
        assert H == self.input_h and W == self.input_w
        self._ensure(B, T)
        dev = self.device
        xd = x.to(dev, torch.float32).contiguous()
        td = t.to(dev, torch.int64).contiguous()
        ad = external_cond.to(dev, torch.float32).contiguous() if torch.is_tensor(external_cond) else None
        out = torch.empty_like(xd)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().gtav_dit_train_forward(self._handle, xd.data_ptr(), td.data_ptr(), _lib.ptr(ad), out.data_ptr(), B, T,
                                                          _lib.current_stream()))
        return out

    def backward_(self, v_pred: torch.Tensor, v_target: torch.Tensor):
        """Adds d mean((v_pred[:, -1] - v_target)^2) / d theta (times the loss scale) to the gradient arena (`accelerator.backward`)."""
        vt = v_target.to(self.device, torch.float32).contiguous()
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().gtav_dit_train_backward(self._handle, v_pred.data_ptr(), vt.data_ptr(), _lib.current_stream()))

    def backward_phases_(self, v_pred: torch.Tensor, v_target: torch.Tensor, phase_begin: int, phase_end: int):
        """Phases [phase_begin, phase_end) of backward_: 0 = loss + final layer, p in 1..depth = block depth - p, depth + 1 = embedders."""
        vt = v_target.to(self.device, torch.float32).contiguous()
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().gtav_dit_train_backward_phases(self._handle, v_pred.data_ptr(), vt.data_ptr(), phase_begin, phase_end,
                                                                 _lib.current_stream()))

    def param_range(self, prefix: str):
        """(offset, count) of the gradient-arena slice of the parameters whose names start with `prefix` (lexicographic layout)."""
        off, cnt = C.c_int64(0), C.c_int64(0)
        _lib.check(_lib.load().gtav_dit_train_param_range(self._handle, prefix.encode(), C.byref(off), C.byref(cnt)))
        return off.value, cnt.value

    def zero_grad(self):
        if not self._handle:
            self._ensure(self._capacity_b, self._capacity_t)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().gtav_dit_zero_grad(self._handle, _lib.current_stream()))

    @property
    def grad_arena(self) -> torch.Tensor:
        """All gradients (loss-scaled), contiguous, parameters in lexicographic name order: what a data-parallel run all-reduces."""
        return self._grads

    @property
    def loss_scale(self) -> float:
        return self._loss_scale

    @loss_scale.setter
    def loss_scale(self, v: float):
        self._loss_scale = float(v)
        if self._handle:
            _lib.check(_lib.load().gtav_dit_set_loss_scale(self._handle, self._loss_scale))

    def grad(self, name: str) -> torch.Tensor:
        """Unscaled gradient of one parameter in its state-dict shape."""
        shp = self._shapes()[name]
        out = torch.empty(shp, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().gtav_dit_get_grad(self._handle, name.encode(), out.data_ptr(), out.numel(), _lib.current_stream()))
        return out / (self._loss_scale * self.grad_divisor)

    def residual_after(self, k: int, B: int, T: int) -> torch.Tensor:
        """Residual stream of the last forward_train after k branch additions (4 per block; k = 4 (i + 1) is the output of block i),
        as (B, T, h, w, D) like the reference's block outputs."""
        gh, gw = self.input_h // self.patch_size, self.input_w // self.patch_size
        out = torch.empty(B * T * gh * gw, self.hidden_size, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().gtav_dit_train_get_residual(self._handle, k, out.data_ptr(), out.numel(), _lib.current_stream()))
        return out.reshape(B, T, gh, gw, self.hidden_size)

    def adamw_step(self, lr: float, weight_decay: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-7, max_grad_norm: float = 0.0):
        """clip_grad_norm_ + AdamW.step on the GPU masters (train_dit.py:232-238, 965-968); refreshes the fp16 GEMM operands."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().gtav_dit_adamw_step(self._handle, float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                                                       float(max_grad_norm), _lib.current_stream()))

    def train_stats(self):
        """(step applied?, skipped steps so far, unscaled global gradient norm of the last step); synchronises."""
        buf = (C.c_float * 4)()
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().gtav_dit_train_stats(self._handle, buf, _lib.current_stream()))
        return buf[1] != 0.0, int(buf[2]), float(buf[3])

    def opt_state_dict(self) -> Dict[str, torch.Tensor]:
        """AdamW state of every trainable parameter as CPU tensors: "m.<name>" / "v.<name>" (first / second moments, state-dict shapes)
        plus "step" = [applied steps, skipped steps] — what accelerator.save_state keeps for the optimizer (train_dit.py:765-800)."""
        assert self._trainable and self._handle, "no optimizer state: construct with trainable=True and run a step (or reserve()) first"
        L = _lib.load()
        out = {}
        with torch.cuda.device(self.device):
            for k, shp in self._shapes().items():
                m = torch.empty(shp, device=self.device, dtype=torch.float32)
                v = torch.empty_like(m)
                _lib.check(L.gtav_dit_get_opt_state(self._handle, k.encode(), m.data_ptr(), v.data_ptr(), m.numel(), _lib.current_stream()))
                out["m." + k], out["v." + k] = m.cpu(), v.cpu()
            a, sk = C.c_int64(0), C.c_int64(0)
            _lib.check(L.gtav_dit_get_opt_step(self._handle, C.byref(a), C.byref(sk), _lib.current_stream()))
        out["step"] = torch.tensor([a.value, sk.value], dtype=torch.int64)
        return out

    def load_opt_state_dict(self, state: Dict[str, torch.Tensor]):
        """Inverse of opt_state_dict (the weights themselves: load_state_dict).  The handle is built first if it does not exist yet."""
        assert self._trainable, "construct the model with trainable=True"
        self._ensure(self._capacity_b, self._capacity_t)
        L = _lib.load()
        with torch.cuda.device(self.device):
            for k, shp in self._shapes().items():
                m = state["m." + k].to(self.device, torch.float32).contiguous()
                v = state["v." + k].to(self.device, torch.float32).contiguous()
                if tuple(m.shape) != tuple(shp) or tuple(v.shape) != tuple(shp):
                    raise RuntimeError(f"load_opt_state_dict: {k} has shape {tuple(m.shape)}, expected {tuple(shp)}")
                _lib.check(L.gtav_dit_set_opt_state(self._handle, k.encode(), m.data_ptr(), v.data_ptr(), m.numel(), _lib.current_stream()))
            st = state["step"]
            _lib.check(L.gtav_dit_set_opt_step(self._handle, int(st[0]), int(st[1]), _lib.current_stream()))
            torch.cuda.synchronize()

    @property
    def grad_divisor(self) -> float:
        return getattr(self, "_grad_divisor", 1.0)

    @grad_divisor.setter
    def grad_divisor(self, v: float):
        """The gradient arena holds the SUM over this many ranks (all-reduce SUM): the optimizer step divides, the arena is not rescaled."""
        self._grad_divisor = float(v)
        if self._handle:
            _lib.check(_lib.load().gtav_dit_set_grad_divisor(self._handle, self._grad_divisor))

    def pull_weights(self):
        """Copies the trained fp32 masters from the GPU back into the host state dict (checkpoints, state_dict())."""
        L = _lib.load()
        with torch.cuda.device(self.device):
            for k, shp in self._shapes().items():
                out = torch.empty(shp, device=self.device, dtype=torch.float32)
                _lib.check(L.gtav_dit_get_weight(self._handle, k.encode(), out.data_ptr(), out.numel(), _lib.current_stream()))
                self._sd[k] = out.cpu()

    # ------------------------------------------------------------------------------------------
    # fused sampler entry (train_dit.denoise_step + generate.py:220), used by gtav_amd.generate
    # ------------------------------------------------------------------------------------------
    def set_schedule(self, alphas_cumprod: torch.Tensor):
        ac = alphas_cumprod.detach().reshape(-1).float().cpu().contiguous()
        assert ac.numel() == 1000
        self._schedule = ac
        if self._handle:
            arr = (C.c_float * 1000).from_buffer_copy(ac.numpy().tobytes())
            _lib.check(_lib.load().gtav_dit_set_schedule(self._handle, arr, 1000))

    def prepare_frame_(self, B: int, F: int, start: int, cur: int, t_ctx: int, t_steps, actions: Optional[torch.Tensor] = None):
        """Builds the conditioning (adaLN) table for every noise step of one generated frame (gtav_dit_prepare_frame);
        step k of that frame then passes cond_step=k to denoise_step_."""
        n = len(t_steps)
        self._ensure(B, cur - start + 1, cond_rows=B * (cur - start + n))
        arr = (C.c_int32 * n)(*[int(v) for v in t_steps])
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().gtav_dit_prepare_frame(self._handle, B, F, start, cur, int(t_ctx), arr, n, _lib.ptr(actions),
                                                          _lib.current_stream()))

    def denoise_step_(self, x: torch.Tensor, start: int, cur: int, t_ctx: int, t_cur: int, t_next: int, is_final: bool,
                      actions: Optional[torch.Tensor] = None, cached: bool = False, v_out: Optional[torch.Tensor] = None,
                      cond_step: int = -1):
        """In-place fused step on latents x (B, F, C, H, W) fp32 contiguous on the model's device."""
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        B, F = x.shape[:2]
        if cond_step < 0:
            self._ensure(B, cur - start + 1)
        with torch.cuda.device(self.device):
            _lib.check(_lib.load().gtav_dit_denoise_step(
                self._handle, x.data_ptr(), B, F, start, cur, int(t_ctx), int(t_cur), int(t_next), int(bool(is_final)),
                _lib.ptr(actions), 1 if cached else 0, int(cond_step), _lib.ptr(v_out), _lib.current_stream()))

    PROFILE_CLASSES = ("ln_modulate", "gemm_qkv", "attn_spatial", "attn_temporal", "gemm_out", "gemm_fc1", "gemm_fc2", "other",
                       "empty_event_pair")

    def profile(self, enable: bool):
        """In-situ per-kernel-class HIP-event timing (measurement passes only; see include/gtav_amd.h)."""
        _lib.check(_lib.load().gtav_dit_profile(self._handle, int(bool(enable))))

    def profile_read(self):
        ms = (C.c_double * 9)()
        n = (C.c_int64 * 9)()
        _lib.check(_lib.load().gtav_dit_profile_read(self._handle, ms, n))
        return {k: (ms[i], n[i]) for i, k in enumerate(self.PROFILE_CLASSES)}

    def set_graph(self, enable: bool):
        """hipGraph replay of the fused sampler step on/off (on by default)."""
        _lib.check(_lib.load().gtav_dit_set_graph(self._handle, int(bool(enable))))

    def set_fused_temporal(self, enable: bool):
        """Temporal QKV projection + temporal attention as one kernel on batch-1 full-window steps (off by default; bit-identical to the two-kernel path;
        1-2 % slower per eager forward, 1-1.6 % faster per replayed captured step: generate.tune_weight_prefetch times both and keeps the faster)."""
        self._fused_temporal = bool(enable)
        if self._handle:
            _lib.check(_lib.load().gtav_dit_set_fused_temporal(self._handle, int(self._fused_temporal)))

    def set_fused_spatial(self, enable: bool):
        """Spatial QKV projection + spatial attention as one kernel on steps of 5 or more frames of 144 tokens (gtav_dit_set_fused_spatial; ON by default where
        the geometry allows; bit-identical to the two-kernel path; generate.tune_weight_prefetch times both on the shape it tunes and keeps the faster)."""
        self._fused_spatial = bool(enable)
        if self._handle:
            _lib.check(_lib.load().gtav_dit_set_fused_spatial(self._handle, int(self._fused_spatial)))

    def fused_launches(self, B: int, T: int, t0: int = 0) -> int:
        """Bit 0: a step of B x T frames runs the fused spatial to_qkv + attention launch on this handle as it is now; bit 1: the fused temporal one
        (gtav_dit_fused_launches).  The profiler books those launches under attn_spatial / attn_temporal."""
        self._ensure(B, T)
        return int(_lib.load().gtav_dit_fused_launches(self._handle, int(B), int(T), int(t0)))

    def set_weight_prefetch(self, mode):
        """L2 prefetch of the next GEMM's weight at small token counts (gtav_dit_set_weight_prefetch; on by default, bit-identical results under every
        mode): False / 0 off, True / 1 every weight, or a per-class word from gtav_amd.generate.prefetch_mode."""
        mode = int(mode)
        if mode not in (0, 1) and (mode >> 16) != 1:
            raise ValueError(f"set_weight_prefetch: mode {mode:#x} is neither 0, 1 nor a per-class word (gtav_amd.generate.prefetch_mode)")
        self._weight_prefetch = mode
        if self._handle:
            _lib.check(_lib.load().gtav_dit_set_weight_prefetch(self._handle, mode))

    def set_fold(self, mode: int, min_tokens_a: int = -1, min_tokens_b: int = -1):
        """EXPERIMENTS BUILD ONLY (gtav_amd.lib.load_experiments(), tools/fold_bench.py): the LayerNorm fold of round 3 (gtav_dit_set_fold, csrc/experiments.h) —
        0 = separate LayerNorm launches everywhere, 1 = folded into the GEMM epilogues from min_tokens_* tokens on, 2 = every seam at every size.
        Correct, and measured slower than the LayerNorm launches at every size: the product library does not carry it."""
        if not getattr(_lib.load(), "_gtav_experiments", False):
            raise _lib.GtavError("set_fold: the LayerNorm fold exists only in the experiments build (gtav_amd.lib.load_experiments())")
        self._fold = (int(mode), int(min_tokens_a), int(min_tokens_b))
        if self._handle:
            _lib.check(_lib.load().gtav_dit_set_fold(self._handle, *self._fold))

    # ------------------------------------------------------------------------------------------
    # operand type / range safety (include/gtav_amd.h "operand type"; reference: bf16 autocast, generate.py:125-127, train_dit.py:190-198)
    # ------------------------------------------------------------------------------------------
    def set_operand_dtype(self, dtype, groups=None):
        """torch.float16 (default: 6-9e-4 relative L2 per forward against the fp32 reference, range +-65504) or torch.bfloat16 (the reference's own autocast type:
        fp32 range, ~8e-3) for the 2-byte GEMM / attention operands of `groups` (None = every layer group; group numbering: n_operand_groups).  The weights of
        the changed groups are converted again from the fp32 host copies at the next call."""
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError(f"set_operand_dtype: {dtype} (torch.float16 or torch.bfloat16)")
        if self._trainable and dtype == torch.bfloat16:
            raise _lib.GtavError("set_operand_dtype: a trainable DiT keeps fp16 operands (its backward pass and loss scaling are fp16)")
        gs = set(range(self.n_operand_groups)) if groups is None else {int(g) for g in groups}
        if any(g < 0 or g >= self.n_operand_groups for g in gs):
            raise ValueError(f"set_operand_dtype: groups {sorted(gs)} outside [0, {self.n_operand_groups})")
        new = (self._bf16_groups | gs) if dtype == torch.bfloat16 else (self._bf16_groups - gs)
        if new == self._bf16_groups:
            return
        if self._handle:
            L = _lib.load()
            for g in sorted(new ^ self._bf16_groups):
                _lib.check(L.gtav_dit_set_operand_dtype(self._handle, g, 1 if g in new else 0))
            self._dirty = True               # the changed groups' weight images are stale: the next call re-sends the weights and finalizes
        self._bf16_groups = new

    def operand_dtypes(self):
        """torch dtype of every operand group, in group order."""
        return [torch.bfloat16 if g in self._bf16_groups else torch.float16 for g in range(self.n_operand_groups)]

    def check(self):
        """Raises if a timestep was out of range, an input held NaN / inf, or an fp16 activation saturated since the last call (synchronises).
        range_policy "auto": a saturation moves exactly the layer groups that saturated to bf16 operands (gtav_dit_autorange) and raises GtavRangeSwitch —
        what was computed since the last check is clipped and must be recomputed; the recomputation runs the switched layers in bf16."""
        if not self._handle:
            return
        L = _lib.load()
        with torch.cuda.device(self.device):
            if self.range_policy != "auto" or self._trainable:
                _lib.check(L.gtav_dit_check(self._handle, _lib.current_stream()))
                return
            n = C.c_int32(0)
            _lib.check(L.gtav_dit_autorange(self._handle, C.byref(n), _lib.current_stream()))
            if n.value:
                d = C.c_int32(0)
                new = set()
                for g in range(self.n_operand_groups):
                    _lib.check(L.gtav_dit_get_operand_dtype(self._handle, g, C.byref(d)))
                    if d.value == 1:
                        new.add(g)
                switched = sorted(new - self._bf16_groups)
                self._bf16_groups = new
                self._dirty = True
                raise _lib.GtavRangeSwitch(f"DiT: an activation exceeded the fp16 range (|x| > 65504) in operand group(s) {switched}; those layers now run "
                                           "on bf16 operands (the reference's own autocast type): results since the last check are clipped — recompute them", switched)


def DiT_S_2(**kwargs):
    """model/dit.py:379-389."""
    return DiT(input_h=18, input_w=32, patch_size=2, hidden_size=1024, depth=16, num_heads=16, max_frames=5, **kwargs)


DiT_models = {"DiT-S/2": DiT_S_2}
