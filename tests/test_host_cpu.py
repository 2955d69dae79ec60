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
    def decls(name):
        hdr = open(os.path.join(ROOT, "include", name)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        return set(re.findall(r"\b(gtav_[a-z0-9_]+)\s*\(", hdr))
    product, hooks = decls("gtav_amd.h"), decls("gtav_amd_testing.h")
    assert hooks == {"gtav_op_gemm_set_stages", "gtav_op_gemm_set_wm"} and not (product & hooks)   # test hooks stay out of the product header
    declared = product | hooks
    assert len(declared) >= 35
    dll = ctypes.CDLL(L.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(dll, n)]
    assert not missing, f"symbols declared in the header but not exported: {missing}"
    # the ctypes binding covers the whole header
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    lib = L.load()
    assert lib.gtav_abi_version() == 4
    assert lib.gtav_last_error() is not None


def test_product_library_has_no_experiment_knobs():
    """the shipped .so must not read the environment and must not export result-corrupting switches."""
    dll = ctypes.CDLL(L.LIB_PATH)
    assert not hasattr(dll, "gtav_op_gemm_set_debug")
    nm = subprocess.run(["nm", "-D", "--undefined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in nm, "libgtav_amd.so imports getenv: an environment knob leaked into the product build"
    for src in ("gemm.hip", "api.hip", "api_dit.hip", "api_train.hip", "api_vae.hip", "api_internal.h", "attention.hip", "attn_tile.h", "elementwise.hip", "skinny.hip"):
        text = open(os.path.join(ROOT, "ai-generated-gtav_amd", "csrc", src)).read()
        assert "getenv(" not in text, f"{src}: raw getenv outside GTAV_ENV_INT"


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
from gtav_amd.generate import all_gather_latents, shard_batch, shard_inputs
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
B = 4
lo, hi = shard_batch(B, rank, world)
# every sample's latents are a function of its GLOBAL id only -> gathered result is sharding-invariant
full = torch.stack([torch.full((3, 2, 4, 4), float(g)) + torch.arange(32.).reshape(2, 4, 4) for g in range(B)])
mine = full[lo:hi].clone()
out = all_gather_latents(mine)
assert out.shape == full.shape and torch.equal(out, full), rank
# the product's input selection (gtav_amd.generate.shard_inputs): this rank's shard of a global batch of 4, gathered over the
# ranks in rank order, is bit-identical to the world-size-1 batch -> what each sequence sees does not depend on the rank count
geo = dict(n_prompt=2, total_frames=5, frame_hw=(8, 12), latent_hw=(4, 6), latent_ch=3, seed=77)
gids, frames, noise = shard_inputs(B, rank, world, **geo)
assert gids == list(range(lo, hi)) and frames.shape == (hi - lo, 2, 3, 8, 12) and noise.shape == (hi - lo, 3, 3, 4, 6)
ids1, frames1, noise1 = shard_inputs(B, 0, 1, **geo)
assert ids1 == [0, 1, 2, 3]
assert torch.equal(all_gather_latents(frames), frames1) and torch.equal(all_gather_latents(noise), noise1), rank
assert torch.equal(frames, frames1[lo:hi]) and torch.equal(noise, noise1[lo:hi])
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


def test_bench_launcher_spawns_n_ranks_and_fails_cleanly_without_gpus():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start 2 rank processes itself (before touching the GPU) and, on a
    host without 2 GPUs, fail with a non-zero status and one "needs 2 GPUs" message per rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    if torch.cuda.device_count() >= 2:
        pytest.skip("this host has 2 GPUs: the failure path is not reachable")
    assert r.returncode != 0
    assert r.stderr.count("needs 2 GPUs") == 2, r.stderr[-2000:]
    assert "rank 0/2" in r.stderr and "rank 1/2" in r.stderr
    assert r.stdout.strip() == ""          # no JSON line from a failed run


def test_bench_launcher_unit():
    """launch_ranks: ranks see RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 and a common port; a failing rank's status
    is returned and the surviving ranks are stopped."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.parse_args(["--gpus", "8"]).gpus == 8 and bench.parse_args([]).batch_per_gpu == 1
    t0 = __import__("time").time()
    rc = bench.launch_ranks(2, ["--gpus", "2", "--mode", "nonsense"])      # argparse rejects it in every rank: exit status 2
    assert rc == 2 and __import__("time").time() - t0 < 120


def test_video_writer_and_one_hot(tmp_path):
    """§8(f)2 host side: uint8 frames -> Motion-JPEG AVI (RIFF structure, frame count, size, round trip within JPEG error),
    .npy and PNG-directory outputs; the 25-way action one-hot with -1 -> zero row (web_dataset.py:22-38)."""
    import struct
    from gtav_amd.data import actions_to_one_hot, read_avi_mjpeg, write_video
    ramp = (torch.linspace(0, 200, 64)[:, None, None] + torch.linspace(0, 40, 96)[None, :, None] + torch.tensor([0., 10., 20.])).byte()
    frames = torch.stack([ramp + 5 * i for i in range(4)])
    p = write_video(str(tmp_path / "v.avi"), frames, fps=10)
    raw = open(p, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"AVI " and struct.unpack("<I", raw[4:8])[0] == len(raw) - 8
    avih = raw.find(b"avih")
    us_per_frame, _, _, _, total = struct.unpack("<5I", raw[avih + 8:avih + 28])
    w, h = struct.unpack("<2I", raw[avih + 8 + 32:avih + 8 + 40])
    assert (us_per_frame, total, w, h) == (100000, 4, 96, 64) and raw.count(b"00dc") >= 8 and b"MJPG" in raw
    back = read_avi_mjpeg(p)
    assert back.shape == frames.shape and (back.int() - frames.int()).abs().float().mean().item() < 3.0
    assert write_video(str(tmp_path / "v.mp4"), frames).endswith((".mp4", ".avi"))
    import numpy as np
    assert np.array_equal(np.load(write_video(str(tmp_path / "v.npy"), frames)), frames.numpy())
    d = write_video(str(tmp_path / "pngs"), frames)
    assert sorted(os.listdir(d)) == [f"frame_{i:04d}.png" for i in range(4)]
    with pytest.raises(ValueError):
        write_video(str(tmp_path / "bad.avi"), frames.float())
    oh = actions_to_one_hot([-1, 3, 0, 24, -1])
    assert oh.shape == (5, 25) and oh.sum().item() == 3 and oh[1, 3] == 1 and oh[0].sum() == 0


def test_cosine_schedule_matches_transformers():
    import pytest
    import torch
    from gtav_amd.train import cosine_with_min_lr
    try:
        from transformers.optimization import get_cosine_with_min_lr_schedule_with_warmup
    except Exception:
        pytest.skip("transformers scheduler not importable")
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=3e-4)
    sch = get_cosine_with_min_lr_schedule_with_warmup(opt, num_warmup_steps=10, num_training_steps=200, num_cycles=0.25, min_lr=1e-5)
    for step in range(200):
        assert abs(sch.get_last_lr()[0] - cosine_with_min_lr(step, 3e-4, 10, 200, 0.25, 1e-5)) < 1e-10, step
        opt.step()
        sch.step()


def _grad_allreduce_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gtav_amd.train import all_reduce_gradients

    class FakeDit:          # the trainer touches .grad_arena (one contiguous fp32 tensor, the GPU model's gradient arena) and .grad_divisor
        grad_arena = torch.arange(10, dtype=torch.float32) * (rank + 1)
        grad_divisor = 1.0
    d = FakeDit()
    all_reduce_gradients(d, world)
    q.put((rank, d.grad_arena.clone() / d.grad_divisor))     # what the optimizer works on: arena / divisor (gtav_dit_set_grad_divisor)
    dist.destroy_process_group()


def test_gradient_allreduce_averages_over_ranks_gloo():
    """train.all_reduce_gradients (DDP's gradient averaging: ONE all-reduce SUM over the contiguous arena, the division by the world size
    handed to the optimizer as dit.grad_divisor instead of a pass over the arena) on two gloo ranks."""
    import socket
    import torch
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_grad_allreduce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = torch.arange(10, dtype=torch.float32) * 1.5          # mean of 1x and 2x
    assert torch.equal(outs[0], want) and torch.equal(outs[1], want)


def test_oracle_autograd_matches_finite_difference():
    """The training oracle (autograd through ref_cpu.dit_forward) against a central finite difference of the loss in two parameter
    directions on a tiny model: pins that `dit_loss_and_grads` differentiates the function the forward tests pin.  The oracle is fp32
    (the reference's CPU path), so the difference quotient is taken with a step of 1e-2 along a unit direction: agreement to 1 %."""
    import torch
    import gtav_amd.weights as W
    from oracle import ref_cpu as O
    kw = dict(input_h=4, input_w=8, patch_size=2, in_channels=16, hidden_size=128, depth=1, num_heads=2, external_cond_dim=25)
    sd = W.synth_state_dict(W.dit_param_shapes(**kw), seed=2)
    cfg = O.DiTConfig(**kw)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 2, 16, 4, 8, generator=g)
    t = torch.tensor([[10, 500]])
    a = torch.zeros(1, 2, 25)
    a[:, :, 3] = 1
    vt = torch.randn(1, 1, 16, 4, 8, generator=g)
    loss, _, grads = O.dit_loss_and_grads(sd, cfg, x, t, a, vt)

    def loss_at(sd2):
        with torch.no_grad():
            return torch.nn.functional.mse_loss(O.dit_forward(sd2, cfg, x, t, a)[:, -1:], vt).double()
    for k in ("blocks.0.t_mlp.fc1.weight", "blocks.0.s_adaLN_modulation.1.weight", "final_layer.linear.weight"):
        d = grads[k] / grads[k].norm()                  # steepest direction: the largest signal against fp32 rounding of the loss
        eps = 1e-2
        sp, sm = dict(sd), dict(sd)
        sp[k] = sd[k] + eps * d
        sm[k] = sd[k] - eps * d
        fd = float((loss_at(sp) - loss_at(sm)) / (2 * eps))
        an = float((grads[k] * d).sum())
        assert abs(fd - an) <= 1e-2 * abs(an) + 1e-6, (k, fd, an)


def test_prefetch_mode_words():
    """gtav_amd.generate.prefetch_mode: the per-class words of gtav_dit_set_weight_prefetch (include/gtav_amd.h) — 0 = off, 1 = every weight whole, otherwise
    0x10000 | nibbles (out-proj, fc1, fc2, to_qkv from bit 0 up)."""
    from gtav_amd.generate import prefetch_mode, PREFETCH_CLASSES
    assert PREFETCH_CLASSES == ("out", "fc1", "fc2", "qkv")
    assert prefetch_mode((0, 0, 0, 0)) == 0 and prefetch_mode((1, 1, 1, 1)) == 1
    assert prefetch_mode((1, 4, 4, 1)) == 0x11441           # the library's default
    assert prefetch_mode((1, 0, 0, 0)) == 0x10001 and prefetch_mode((0, 0, 0, 2)) == 0x12000
    for cls in [(1, 4, 0, 1), (0, 1, 1, 1), (15, 15, 15, 15)]:
        w = prefetch_mode(cls)
        assert (w >> 16) == 1 and tuple((w >> (4 * c)) & 15 for c in range(4)) == cls
