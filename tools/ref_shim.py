"""Import shim for the UPSTREAM REFERENCE (oracle tooling only — never used at run time).

The reference (`/root/reference`) imports `timm`, `diffusers`, `torchvision`, `wandb` and
`webdataset`, none of which exist in the build container (no network).  This module registers
minimal stand-in modules for exactly the names the hot path touches so that
`model.dit`, `model.vae`, `train_dit.denoise_step` and `utils.sigmoid_beta_schedule` can be imported
on CPU and used to (a) validate `oracle/ref_cpu.py` and (b) generate the golden vectors committed
under `tests/golden/` (see `tools/make_golden.py`).

Only third-party *library* surfaces are stubbed (their arithmetic is restated from their published
semantics: timm `Mlp` = fc1 -> act -> drop -> norm(Identity) -> fc2 -> drop).  No reference source is
copied; nothing is written into `/root/reference` (run with PYTHONDONTWRITEBYTECODE=1).

Nothing under `tests/`, `bench.py` or `__graft_entry__` imports this file: `/root/reference` does not
exist on the GPU box.
"""
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("GTAV_REFERENCE_ROOT", "/root/reference")


def _mod(name):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    m.__path__ = []  # behave as a package so that submodule imports resolve through sys.modules
    sys.modules[name] = m
    return m


def install():
    """Register the stand-in modules and put the reference on sys.path. Idempotent."""
    if getattr(install, "_done", False):
        return
    sys.dont_write_bytecode = True
    import torch
    from torch import nn

    # --- timm: Mlp + to_2tuple (used at model/dit.py:14-15, model/vae.py:14-15) -------------
    class Mlp(nn.Module):
        """timm.models.vision_transformer.Mlp semantics (fc1, act, drop1, norm, fc2, drop2)."""

        def __init__(self, in_features, hidden_features=None, out_features=None,
                     act_layer=nn.GELU, norm_layer=None, bias=True, drop=0.0, use_conv=False):
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
            self.act = act_layer()
            self.drop1 = nn.Dropout(drop)
            self.norm = norm_layer(hidden_features) if norm_layer is not None else nn.Identity()
            self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
            self.drop2 = nn.Dropout(drop)

        def forward(self, x):
            return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    timm = _mod("timm")
    timm_models = _mod("timm.models")
    timm_vt = _mod("timm.models.vision_transformer")
    timm_layers = _mod("timm.layers")
    timm_helpers = _mod("timm.layers.helpers")
    timm_vt.Mlp = Mlp
    timm_helpers.to_2tuple = to_2tuple
    timm.models, timm.layers = timm_models, timm_layers
    timm_models.vision_transformer = timm_vt
    timm_layers.helpers = timm_helpers

    # --- diffusers: imported by model/embeddings.py:11, never instantiated on the path -------
    diffusers = _mod("diffusers")
    d_models = _mod("diffusers.models")
    d_emb = _mod("diffusers.models.embeddings")

    class TimestepEmbedding(nn.Module):  # placeholder: dead code on the hot path
        def __init__(self, *a, **k):
            raise RuntimeError("diffusers TimestepEmbedding is not on the hot path")

    d_emb.TimestepEmbedding = TimestepEmbedding
    diffusers.models = d_models
    d_models.embeddings = d_emb

    # --- torchvision / wandb / webdataset / matplotlib: I/O only, import-time names ----------
    tv = _mod("torchvision")
    tv_io = _mod("torchvision.io")
    tv_tf = _mod("torchvision.transforms")
    tv_ut = _mod("torchvision.utils")

    def _unavailable(*a, **k):
        raise RuntimeError("I/O helper not available in the oracle shim")

    tv_io.write_video = tv_io.read_image = _unavailable
    tv_tf.Compose = tv_tf.ToTensor = tv_tf.Resize = _unavailable
    tv_ut.make_grid = _unavailable
    tv.io, tv.transforms, tv.utils = tv_io, tv_tf, tv_ut
    wandb = _mod("wandb")
    wandb.run = None
    _mod("webdataset")
    try:
        import matplotlib  # noqa: F401
    except Exception:
        mpl = _mod("matplotlib")
        mpl.pyplot = _mod("matplotlib.pyplot")

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    install._done = True


def import_reference():
    """Returns (dit_module, vae_module, denoise_step, sigmoid_beta_schedule)."""
    install()
    import model.dit as ref_dit
    import model.vae as ref_vae
    from train_dit import denoise_step
    from utils import sigmoid_beta_schedule
    return ref_dit, ref_vae, denoise_step, sigmoid_beta_schedule
