"""CPU (no GPU): the C-ABI library loads and exports every symbol include/gtav_amd.h declares, the host-side
layout logic matches the reference's state-dict, and the multi-rank path (batch sharding + all-gather) is
correct under gloo with world_size 2."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import gtav_amd.weights as W  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


def test_cabi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gtav_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gtav_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 35
    dll = ctypes.CDLL(L.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(dll, n)]
    assert not missing, f"symbols declared in the header but not exported: {missing}"
    # the ctypes binding covers the whole header
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    lib = L.load()
    assert lib.gtav_abi_version() == 1
    assert lib.gtav_last_error() is not None


def test_cabi_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device (no compute without a GPU)."""
    lib = L.load()
    assert lib.gtav_dit_create(None, None) != 0
    assert b"null argument" in lib.gtav_last_error()
    cfg = L.DitConfig(input_h=18, input_w=32, patch_size=2, in_channels=16, hidden_size=1000, depth=2, num_heads=16, mlp_ratio=4.0,
                      external_cond_dim=25, max_frames=5, max_batch=1, max_cond_rows=5)
    h = ctypes.c_void_p()
    assert lib.gtav_dit_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"hidden_size" in lib.gtav_last_error()
    cfg.hidden_size, cfg.num_heads = 1024, 8
    assert lib.gtav_dit_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"head_dim 64" in lib.gtav_last_error()


def test_state_dict_layout_matches_reference_probe():
    """DiT-S/2 has 607 943 792 parameters (334 entries incl. 34 freqs aliases), the VAE 229 246 160 (228)."""
    s = W.dit_param_shapes(depth=16)
    n = sum(int(torch.Size(v).numel()) for v in s.values())
    assert n + 16 + 32 == 607_943_792          # + the two shared rotary freqs parameters
    assert len(s) + len(list(W.dit_freq_alias_names(16))) == 334
    v = W.vae_param_shapes()
    assert sum(int(torch.Size(x).numel()) for x in v.values()) == 229_246_160 and len(v) == 228
    assert s["blocks.3.t_attn.to_qkv.weight"] == (3072, 1024) and s["x_embedder.proj.weight"] == (1024, 16, 2, 2)
    assert v["predictor.weight"] == (1200, 1024) and v["quant_conv.weight"] == (32, 1024)


def test_synthetic_weights_are_deterministic_and_nonzero_adaln():
    a = W.synth_tensor("blocks.0.s_adaLN_modulation.1.weight", (12, 8), seed=0)
    b = W.synth_tensor("blocks.0.s_adaLN_modulation.1.weight", (12, 8), seed=0)
    c = W.synth_tensor("blocks.0.s_adaLN_modulation.1.weight", (12, 8), seed=1)
    assert torch.equal(a, b) and not torch.equal(a, c) and a.abs().min() > 0


def test_freq_alias_handling(tmp_path):
    """Both on-disk alias conventions of the shared rotary freqs and neither are accepted."""
    kw = dict(input_h=8, input_w=16, hidden_size=256, depth=1, num_heads=4)
    sd = W.synth_state_dict(W.dit_param_shapes(**kw), seed=0)
    f_s, f_t = W.rope_freqs_pixel(32, 256), W.rope_freqs_lang(64)
    for extra in ({}, {"spatial_rotary_emb.freqs": f_s * 2, "temporal_rotary_emb.freqs": f_t},
                  {"blocks.0.s_attn.rotary_emb.freqs": f_s * 2, "blocks.0.t_attn.rotary_emb.freqs": f_t}):
        full = dict(sd, **extra)
        p = str(tmp_path / "m.safetensors")
        W.save_state_dict_file(full, p)
        params, sf, tf = W.split_freq_keys(W.load_state_dict_file(p))
        assert set(params) == set(sd)
        if extra:
            assert torch.equal(sf, f_s * 2) and torch.equal(tf, f_t)
        else:
            assert sf is None and tf is None


def test_model_classes_mirror_reference_signatures():
    import inspect
    from gtav_amd.model.dit import DiT, DiT_models
    from gtav_amd.model.vae import AutoencoderKL, VAE_models
    p = list(inspect.signature(DiT.__init__).parameters)
    assert p[1:11] == ["input_h", "input_w", "patch_size", "in_channels", "hidden_size", "depth", "num_heads", "mlp_ratio",
                       "external_cond_dim", "max_frames"]
    d = inspect.signature(DiT.__init__).parameters
    assert (d["depth"].default, d["hidden_size"].default, d["external_cond_dim"].default, d["max_frames"].default) == (12, 1024, 25, 5)
    v = list(inspect.signature(AutoencoderKL.__init__).parameters)
    assert v[1:14] == ["latent_dim", "input_height", "input_width", "patch_size", "enc_dim", "enc_depth", "enc_heads", "dec_dim",
                       "dec_depth", "dec_heads", "mlp_ratio", "norm_layer", "use_variational"]
    assert set(DiT_models) == {"DiT-S/2"} and set(VAE_models) == {"vit-l-20-shallow-encoder"}
    m = DiT(input_h=8, input_w=16, hidden_size=256, depth=1, num_heads=4, init_weights=False)
    m.max_frames = 7            # assigned by generate.py:139
    assert m.max_frames == 7 and m.patch_size == 2
    with pytest.raises(RuntimeError):
        m.load_state_dict({"bogus": torch.zeros(1)})


def test_shard_batch():
    from gtav_amd.generate import shard_batch
    assert [shard_batch(64, r, 8) for r in (0, 7)] == [(0, 8), (56, 64)]
    with pytest.raises(AssertionError):
        shard_batch(10, 0, 4)


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from gtav_amd.generate import all_gather_latents, shard_batch
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
B = 4
lo, hi = shard_batch(B, rank, world)
# every sample's latents are a function of its GLOBAL id only -> gathered result is sharding-invariant
full = torch.stack([torch.full((3, 2, 4, 4), float(g)) + torch.arange(32.).reshape(2, 4, 4) for g in range(B)])
mine = full[lo:hi].clone()
out = all_gather_latents(mine)
assert out.shape == full.shape and torch.equal(out, full), rank
dist.barrier()
if rank == 0:
    print("GATHER_OK")
dist.destroy_process_group()
"""


def test_all_gather_world_size_2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29611", str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "GATHER_OK" in r.stdout
