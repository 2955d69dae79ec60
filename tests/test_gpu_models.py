"""Model-level parity (GPU): the HIP path behind the reference's class API vs the CPU oracle (oracle/ref_cpu.py, itself
pinned to the reference by tests/golden) on identical seeded inputs and weights.

Tolerance: the north-star bound is 1e-3 relative L2 per forward at full size (fp16 MFMA operands, fp32 accumulate /
residual / LN / softmax, exact-fp32 conditioning path).  Small configs get 2e-3 headroom."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import dev  # noqa: E402
from helpers import rel_l2 as _rel_l2  # noqa: E402
from oracle import ref_cpu as O  # noqa: E402
import gtav_amd.weights as W  # noqa: E402
from gtav_amd.model.dit import DiT, DiT_models  # noqa: E402
from gtav_amd.model.vae import AutoencoderKL, VAE_models  # noqa: E402

TOL_FULL = 1e-3      # north-star bound: full-size DiT / VAE forwards vs the fp32 CPU reference
TOL_SMALL = 2e-3     # toy widths (hidden 128-256): fewer terms per dot product average the fp16 operand rounding less
TOL_ROLLOUT = 1.5e-3   # tens of chained forwards (toy and full-size models; measured <= 7e-4, pytest -s prints every margin)


def rel_l2(a, b):
    v = _rel_l2(a, b)
    print(f"[rel_l2 {os.environ.get('PYTEST_CURRENT_TEST', '').split('::')[-1].split(' ')[0]}] {v:.3e}")
    return v


SMALL_DIT = dict(input_h=8, input_w=16, patch_size=2, in_channels=16, hidden_size=256, depth=2, num_heads=4, external_cond_dim=25)
SMALL_VAE = dict(latent_dim=16, input_height=64, input_width=96, patch_size=8, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=256,
                 dec_depth=2, dec_heads=4)


def _mk_dit(kw, seed, max_batch=2):
    sd = W.synth_state_dict(W.dit_param_shapes(**kw), seed=seed)
    m = DiT(**kw, max_batch=max_batch, init_weights=False)
    m.load_state_dict(sd)
    return m, sd, O.DiTConfig(**kw)


def _inputs(cfg, B, T, seed, actions=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, cfg.in_channels, cfg.input_h, cfg.input_w, generator=g)
    t = torch.randint(0, 1000, (B, T), generator=g)
    a = None
    if actions:
        a = torch.zeros(B, T, 25)
        a[torch.arange(B)[:, None], torch.arange(T)[None], torch.randint(0, 25, (B, T), generator=g)] = 1
    return x, t, a


@pytest.mark.parametrize("actions", [False, True])
def test_small_dit_forward(actions):
    m, sd, cfg = _mk_dit(SMALL_DIT, seed=3)
    x, t, a = _inputs(cfg, 2, 3, seed=11, actions=actions)
    with torch.no_grad():
        ref = O.dit_forward(sd, cfg, x, t, a)
    out = m(x, t, a)
    assert out.shape == ref.shape and out.dtype == torch.float32
    assert rel_l2(out, ref) < TOL_SMALL


@pytest.fixture(scope="module")
def full_dit():
    m = DiT_models["DiT-S/2"](init_weights=False, max_batch=2)
    sd = W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0)
    m.load_state_dict(sd)
    return m, sd, O.dit_s_2()


def test_full_dit_forward_b1_t5(full_dit):
    m, sd, cfg = full_dit
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 5, 16, 18, 32, generator=g)
    t = torch.tensor([[15, 15, 15, 15, 500]])
    a = torch.zeros(1, 5, 25)
    a[:, :, 3] = 1
    with torch.no_grad():
        ref = O.dit_forward(sd, cfg, x, t, a)
    out = m(x, t, a)
    e = rel_l2(out, ref)
    print("full DiT B=1 T=5 rel-L2", e)
    assert e < 1e-3


def test_full_dit_forward_b2_t3_no_actions(full_dit):
    m, sd, cfg = full_dit
    x, t, _ = _inputs(cfg, 2, 3, seed=5, actions=False)
    with torch.no_grad():
        ref = O.dit_forward(sd, cfg, x, t, None)
    e = rel_l2(m(x, t, None), ref)
    print("full DiT B=2 T=3 rel-L2", e)
    assert e < 1e-3


def test_fused_temporal_qkv_attention_is_bit_identical(full_dit):
    """gtav_dit_set_fused_temporal(h, 1): batch-1 five-frame windows run the temporal to_qkv projection + temporal attention as
    one kernel (gemm_qkvt_attn_kernel); every other shape, and the default, the two-kernel path.  Same fp16
    operands and the same arithmetic order: the outputs must be EQUAL, at full size (144 blocks) and on a toy model with two
    batch items (tile -> (batch item, position group) indexing), through the plain forward and the captured sampler step."""
    from gtav_amd.utils import alphas_cumprod
    m, sd, cfg = full_dit
    g = torch.Generator().manual_seed(21)
    x = torch.randn(1, 5, 16, 18, 32, generator=g)
    t = torch.tensor([[15, 15, 15, 15, 700]])
    a = torch.zeros(1, 5, 25)
    a[:, :, 7] = 1
    try:
        m.set_fused_temporal(False)
        split = m(x, t, a).clone()
        m.set_fused_temporal(True)
        fused = m(x, t, a).clone()
    finally:
        m.set_fused_temporal(False)
    assert torch.isfinite(fused).all() and torch.equal(fused, split)
    with torch.no_grad():
        assert rel_l2(fused, O.dit_forward(sd, cfg, x, t, a)) < TOL_FULL
    ms, _, _ = _mk_dit(SMALL_DIT, seed=9)
    xs, ts, as_ = _inputs(O.DiTConfig(**SMALL_DIT), 2, 5, seed=13)
    ms.set_fused_temporal(False)
    split = ms(xs, ts, as_).clone()
    ms.set_fused_temporal(True)
    assert torch.equal(ms(xs, ts, as_), split)
    # sampler steps (eager warm-up, capture, replay) and a context-cached step on the K/V cache the fused kernel wrote
    ms.set_schedule(alphas_cumprod(1e-4))
    outs = []
    for fused_on in (False, True):
        ms.set_fused_temporal(fused_on)
        xd = xs.to(dev()).contiguous()
        for _ in range(3):
            ms.denoise_step_(xd, 0, 4, 15, 600, 500, False, as_.to(dev()))
        ms.denoise_step_(xd, 0, 4, 15, 500, 400, False, as_.to(dev()), cached=True)
        outs.append(xd.clone())
    assert torch.equal(outs[0], outs[1])



def test_fused_spatial_qkv_attention_is_bit_identical(full_dit):
    """gtav_dit_set_fused_spatial(h, 1): steps / forwards of 4 ... 16 frames of 144 tokens run the spatial to_qkv projection + spatial attention as one kernel
    (gemm_qkvs_attn_kernel); every other shape, and the default, the two-kernel path.  Same fp16 operands, the same tile body: the outputs must be EQUAL —
    batch 1 and 2 through the plain forward, alone and together with the fused temporal launch, and through the sampler (eager, captured, replayed, then a
    context-cached step, which is too small for the fused kernel and runs the split path on the same caches)."""
    from gtav_amd.utils import alphas_cumprod
    m, sd, cfg = full_dit
    g = torch.Generator().manual_seed(23)
    a = torch.zeros(2, 5, 25)
    a[:, :, 7] = 1
    x = torch.randn(2, 5, 16, 18, 32, generator=g)
    t = torch.tensor([[15, 15, 15, 15, 700], [15, 15, 15, 15, 320]])
    try:
        m.set_fused_spatial(False)
        m.set_fused_temporal(False)
        assert m.fused_launches(1, 5) == 0
        split1, split2 = m(x[:1], t[:1], a[:1]).clone(), m(x, t, a).clone()
        m.set_fused_spatial(True)
        fused1, fused2 = m(x[:1], t[:1], a[:1]).clone(), m(x, t, a).clone()
        assert m.fused_launches(1, 5) == 1 and m.fused_launches(2, 5) == 1 and m.fused_launches(1, 1) == 0 and m.fused_launches(8, 1) == 1
        m.profile(True)                                   # the profiler books a fused launch under the attention class of its half: one class, one kernel
        m(x[:1], t[:1], a[:1])
        pr = m.profile_read()
        m.profile(False)
        assert pr["gemm_qkv"][1] == cfg.depth and pr["attn_spatial"][1] == cfg.depth and pr["attn_temporal"][1] == cfg.depth
        m.set_fused_temporal(True)
        assert m.fused_launches(1, 5) == 3 and m.fused_launches(2, 5) == 1
        both1 = m(x[:1], t[:1], a[:1]).clone()
        assert torch.isfinite(fused1).all() and torch.equal(fused1, split1) and torch.equal(fused2, split2) and torch.equal(both1, split1)
        with torch.no_grad():
            assert rel_l2(fused1, O.dit_forward(sd, cfg, x[:1], t[:1], a[:1])) < TOL_FULL
        m.set_fused_temporal(False)
        m.set_schedule(alphas_cumprod(1e-4))
        outs = []
        for fused_on in (False, True):
            m.set_fused_spatial(fused_on)
            xd = x[:1].to(dev()).contiguous()
            for _ in range(3):
                m.denoise_step_(xd, 0, 4, 15, 600, 500, False, a[:1].to(dev()))
            m.denoise_step_(xd, 0, 4, 15, 500, 400, False, a[:1].to(dev()), cached=True)
            outs.append(xd.clone())
        assert torch.equal(outs[0], outs[1])
        m.check()
    finally:
        m.set_fused_spatial(True)        # the library's default on this geometry
        m.set_fused_temporal(False)


def test_weight_prefetch_switch_is_bit_identical_and_tunable(full_dit):
    """gtav_dit_set_weight_prefetch (round 4; per-class modes round 5): the next-weight L2 prefetch of the small-M GEMMs changes no arithmetic — a captured
    batch-1 window step gives EQUAL latents under every mode — and generate.tune_weight_prefetch times the settings, leaves the model on the fastest one and
    reports the choice."""
    from gtav_amd.generate import tune_weight_prefetch, prefetch_mode
    from gtav_amd.utils import alphas_cumprod
    m, _, _ = full_dit
    g = torch.Generator().manual_seed(17)
    x0 = (torch.randn(1, 5, 16, 18, 32, generator=g) * 0.5).to(dev())
    m.set_schedule(alphas_cumprod(1e-4))
    outs = []
    try:
        for mode in (1, 0, prefetch_mode((1, 0, 4, 1)), prefetch_mode((0, 0, 0, 2))):     # on, off, two per-class settings (round 5)
            m.set_weight_prefetch(mode)
            x = x0.clone()
            for k in range(4):                       # eager warm-up, capture, two replays
                m.denoise_step_(x, 0, 4, 15, 900 - 10 * k, 890 - 10 * k, False, None)
            outs.append(x.clone())
        assert all(torch.equal(outs[0], o) for o in outs[1:])
        r = tune_weight_prefetch(m, 1, steps=6, rounds=1)
        assert r["chosen"] in ("on", "off", "per-class") and r["on_ms"] > 0 and r["off_ms"] > 0 and r["tuned_ms"] <= min(r["on_ms"], r["off_ms"])
        assert r["mode"] == prefetch_mode(tuple(r["classes"][c] for c in ("out", "fc1", "fc2", "qkv")))
        print("weight prefetch on / off / tuned (ms per step):", r)
        ft = r["fused_temporal_qkv_attention"]                  # (round 6) the fused temporal to_qkv + attention launch, timed the same way on this five-frame window
        assert ft is not None and ft["chosen"] in ("on", "off") and ft["on_ms"] > 0 and ft["off_ms"] > 0
        fs = r["fused_spatial_qkv_attention"]                   # and the fused spatial launch (frames of 144 tokens, 5 frames)
        assert fs is not None and fs["chosen"] in ("on", "off") and fs["on_ms"] > 0 and fs["off_ms"] > 0
        x = x0.clone()
        for k in range(4):
            m.denoise_step_(x, 0, 4, 15, 900 - 10 * k, 890 - 10 * k, False, None)
        assert torch.equal(x, outs[0])                            # whatever the tuner kept: the same latents
        with pytest.raises(Exception):
            m.set_weight_prefetch(7)                 # neither 0, 1 nor a per-class word
        # the form bench.py uses for the context-cached step of the batched leg: a one-frame window at batch 2 (288 tokens: inside the prefetch's range)
        r1 = tune_weight_prefetch(m, 2, window=1, steps=4, rounds=1)
        assert r1["tuned_ms"] > 0 and set(r1["classes"]) == {"out", "fc1", "fc2", "qkv"}
        assert r1["fused_temporal_qkv_attention"] is None and r1["fused_spatial_qkv_attention"] is None   # 2 frames: neither launch is eligible
    finally:
        m.set_weight_prefetch(True)
        m.set_fused_temporal(False)
        m.set_fused_spatial(True)        # the library's default on this geometry
    m.check()


def test_denoise_step_mirror_and_fused_and_cached():
    from gtav_amd.sampler import denoise_step
    from gtav_amd.utils import alphas_cumprod
    m, sd, cfg = _mk_dit(SMALL_DIT, seed=4)
    g = torch.Generator().manual_seed(2)
    B, n = 2, 6
    x = torch.randn(B, n, 16, 8, 16, generator=g)
    a = torch.zeros(B, n, 25)
    a[:, :, 3] = 1
    ac = alphas_cumprod(1e-4)
    nr = torch.linspace(0, 999, 11)
    dit_fn = lambda xx, tt, aa: O.dit_forward(sd, cfg, xx, tt, aa)
    for noise_idx in (10, 4, 0):
        with torch.no_grad():
            xr, vr = O.denoise_step(dit_fn, x, a, noise_idx, 15, nr, ac[:, None, None, None], start_frame=1)
        xp, vp = denoise_step(m, x, a, noise_idx, 15, nr, ac[:, None, None, None], start_frame=1)
        assert rel_l2(vp, vr) < TOL_SMALL and rel_l2(xp, xr) < TOL_SMALL
        # fused in-place step (window recompute) == mirror on the last frame
        xd = x.to(dev()).contiguous()
        m.set_schedule(ac)
        t_cur, t_next = int(nr[noise_idx]), int(nr[max(0, noise_idx - 1)])
        m.denoise_step_(xd, 1, n - 1, 15, t_cur, t_next, noise_idx <= 0, a.to(dev()))
        assert rel_l2(xd[:, -1], xr[:, -1]) < TOL_SMALL
        assert torch.equal(xd[:, :-1].cpu(), x[:, :-1])
        # context-cached step reproduces the window step (same kernels, same rows).  It needs the K/V caches of a full-window
        # step on the SAME buffer (the handle refuses anything else): run one, restore the frame it updated, then the cached step
        xc = x.to(dev()).contiguous()
        m.denoise_step_(xc, 1, n - 1, 15, t_cur, t_next, noise_idx <= 0, a.to(dev()))
        xc[:, -1] = x[:, -1].to(dev())
        m.denoise_step_(xc, 1, n - 1, 15, t_cur, t_next, noise_idx <= 0, a.to(dev()), cached=True)
        assert rel_l2(xc[:, -1], xd[:, -1]) < 1e-5


def test_small_vae_encode_decode():
    sd = W.synth_state_dict(W.vae_param_shapes(**SMALL_VAE), seed=5)
    v = AutoencoderKL(**SMALL_VAE, init_weights=False)
    v.load_state_dict(sd)
    cfg = O.VAEConfig(**SMALL_VAE)
    g = torch.Generator().manual_seed(1)
    img = torch.rand(3, 3, 64, 96, generator=g) * 2 - 1
    with torch.no_grad():
        mom = O.vae_encode_moments(sd, cfg, img)
    post = v.encode(img)
    assert rel_l2(post.mean, mom[..., :16]) < TOL_SMALL
    assert rel_l2(post.logvar, mom[..., 16:].clamp(-30, 20)) < TOL_SMALL
    z = torch.randn(3, cfg.seq_len, 16, generator=g)
    with torch.no_grad():
        ref = O.vae_decode(sd, cfg, z)
    assert rel_l2(v.decode(z), ref) < TOL_SMALL


def test_full_vae_encode_decode():
    v = VAE_models["vit-l-20-shallow-encoder"](init_weights=False)
    sd = W.synth_state_dict(W.vae_param_shapes(), seed=1)
    v.load_state_dict(sd)
    cfg = O.vit_l_20_shallow_encoder()
    g = torch.Generator().manual_seed(3)
    img = torch.rand(2, 3, 360, 640, generator=g) * 2 - 1
    with torch.no_grad():
        mom = O.vae_encode_moments(sd, cfg, img)
    e1 = rel_l2(v.encode(img).mean, mom[..., :16])
    z = torch.randn(2, 576, 16, generator=g)
    with torch.no_grad():
        ref = O.vae_decode(sd, cfg, z)
    e2 = rel_l2(v.decode(z), ref)
    print("full VAE encode rel-L2", e1, "decode rel-L2", e2)
    assert e1 < TOL_FULL and e2 < TOL_FULL


def test_small_rollout_config1_shape():
    """BASELINE config 1 shape (1 prompt frame -> 4 frames, 10 noise steps => 33 forwards, windows 2/3/4) on the small
    DiT: window-recompute and ctx-cached algorithms vs the oracle rollout with identical injected noise."""
    from gtav_amd.generate import generate_latents
    m, sd, cfg = _mk_dit(SMALL_DIT, seed=6)
    g = torch.Generator().manual_seed(9)
    B = 2
    x0 = torch.randn(B, 1, 16, 8, 16, generator=g) * 0.5
    noise = torch.randn(B, 3, 16, 8, 16, generator=g)
    a = torch.zeros(B, 4, 25)
    a[:, :, 3] = 1
    dit_fn = lambda xx, tt, aa: O.dit_forward(sd, cfg, xx, tt, aa)
    with torch.no_grad():
        ref = O.generate_latents(dit_fn, x0, 4, 10, noise, a, max_frames=5)
    out = generate_latents(m, x0, 4, 10, noise, a)
    out_c = generate_latents(m, x0, 4, 10, noise, a, ctx_cache=True)
    out_i = generate_latents(m, x0, 4, 10, noise, a, hoist_cond=False)   # conditioning recomputed inside every step
    e, ec = rel_l2(out, ref), rel_l2(out_c, ref)
    print("rollout rel-L2 window", e, "cached", ec, "cached-vs-window", rel_l2(out_c, out), "inline-cond-vs-hoisted", rel_l2(out_i, out))
    assert e < TOL_ROLLOUT and ec < TOL_ROLLOUT   # 33 chained forwards; per-forward bound is tested above
    assert rel_l2(out_c, out) < 1e-4
    assert rel_l2(out_i, out) < 1e-6   # hoisting the conditioning is the same arithmetic


def test_train_forward_loss():
    from gtav_amd.train import forward_loss
    m, sd, cfg = _mk_dit(SMALL_DIT, seed=7, max_batch=3)
    g = torch.Generator().manual_seed(4)
    B = 3
    lat = torch.randn(B, 5, 16, 8, 16, generator=g) * 0.5
    a = torch.zeros(B, 5, 25)
    a[:, -1, 1] = 1
    tgt = torch.tensor([50, 1, 23])
    ctx = torch.tensor([40, 7, 30])
    ctx_noise = torch.randn(B, 4, 16, 8, 16, generator=g) * 8   # exercises the +-20 clamp
    noise = torch.randn(B, 1, 16, 8, 16, generator=g) * 8
    dit_fn = lambda xx, tt, aa: O.dit_forward(sd, cfg, xx, tt, aa)
    with torch.no_grad():
        loss_r, vp_r, vt_r, xn_r, t_r = O.train_forward_loss(dit_fn, lat, a, tgt, ctx, ctx_noise, noise)
    loss, vp, vt = forward_loss(m, lat, a, tgt, ctx, ctx_noise, noise)
    assert rel_l2(vt, vt_r) < 1e-6
    assert rel_l2(vp, vp_r) < TOL_SMALL
    assert abs(loss.item() - loss_r.item()) / loss_r.item() < TOL_SMALL


def test_state_dict_round_trip_through_the_cabi(tmp_path):
    """load from a .safetensors file (either alias convention), read every parameter back through gtav_dit_get_weight:
    fp32 conditioning weights exactly, fp16-packed GEMM weights to fp16 rounding."""
    import ctypes as C
    from gtav_amd import lib as L
    from gtav_amd.weights import load_state_dict_file, save_state_dict_file
    kw = dict(SMALL_DIT)
    sd = W.synth_state_dict(W.dit_param_shapes(**kw), seed=9)
    path = str(tmp_path / "dit.safetensors")
    save_state_dict_file(dict(sd, **{"blocks.0.s_attn.rotary_emb.freqs": W.rope_freqs_pixel(32, 256),
                                     "blocks.0.t_attn.rotary_emb.freqs": W.rope_freqs_lang(64)}), path)
    m = DiT(**kw, init_weights=False)
    missing, unexpected = m.load_state_dict(load_state_dict_file(path))
    assert not missing and not unexpected
    x, t, a = _inputs(O.DiTConfig(**kw), 1, 2, seed=1)
    m(x, t, a)   # creates the handle and uploads
    lib = L.load()
    for name, ref in sd.items():
        out = torch.empty(ref.numel(), device=dev())
        L.check(lib.gtav_dit_get_weight(m._handle, name.encode(), out.data_ptr(), ref.numel(), L.current_stream()))
        got = out.cpu().reshape(ref.shape)
        exact = ("adaLN" in name or name.startswith("t_embedder") or name.startswith("external_cond") or name.endswith(".bias"))
        if exact:
            assert torch.equal(got, ref), name
        else:
            assert torch.equal(got, ref.half().float()), name
    assert set(m.state_dict()) == set(sd) | {"spatial_rotary_emb.freqs", "temporal_rotary_emb.freqs"}


def test_dit_edge_shapes_and_errors():
    m, sd, cfg = _mk_dit(SMALL_DIT, seed=3)
    # T = 1 (single frame window) and batch growth beyond the initial capacity
    for B, T in ((1, 1), (3, 2)):
        x, t, a = _inputs(cfg, B, T, seed=20 + B)
        with torch.no_grad():
            ref = O.dit_forward(sd, cfg, x, t, a)
        assert rel_l2(m(x, t, a), ref) < TOL_SMALL
    x, t, a = _inputs(cfg, 1, 2, seed=5)
    with pytest.raises(AssertionError):
        m(x[..., :4, :], t, a)            # wrong spatial size (model/dit.py:67-69)
    bad_t = t.clone()
    bad_t[0, 0] = 1000                    # outside the schedule
    m(x, bad_t, a)
    from gtav_amd.lib import GtavError
    with pytest.raises(GtavError):
        m.check()


def test_large_token_count_paths():
    """A few thousand tokens at small width (small DiT at B=16, T=5 -> 2560 tokens; small VAE on 24 frames): grids of several
    hundred blocks and multi-round LayerNorm / attention launches at hidden 256 / 128.  (The full-size production shapes at
    M = 5760 / 11 520 are covered by test_full_dit_batch8_production_shapes / test_full_dit_batch16_train_forward_loss.)"""
    m, sd, cfg = _mk_dit(SMALL_DIT, seed=8, max_batch=16)
    x, t, a = _inputs(cfg, 16, 5, seed=31)
    with torch.no_grad():
        ref = O.dit_forward(sd, cfg, x, t, a)
    assert rel_l2(m(x, t, a), ref) < TOL_SMALL
    vsd = W.synth_state_dict(W.vae_param_shapes(**SMALL_VAE), seed=5)
    v = AutoencoderKL(**SMALL_VAE, init_weights=False, max_frames_per_call=24)
    v.load_state_dict(vsd)
    vcfg = O.VAEConfig(**SMALL_VAE)
    g = torch.Generator().manual_seed(2)
    img = torch.rand(24, 3, 64, 96, generator=g) * 2 - 1
    with torch.no_grad():
        mom = O.vae_encode_moments(vsd, vcfg, img)
    assert rel_l2(v.encode(img).mean, mom[..., :16]) < TOL_SMALL
    z = torch.randn(24, vcfg.seq_len, 16, generator=g)
    with torch.no_grad():
        ref = O.vae_decode(vsd, vcfg, z)
    assert rel_l2(v.decode(z), ref) < TOL_SMALL


def test_g256_geometry_preset():
    """BASELINE.json says '256x256' frames; the reference's factories are 360x640 only, but its constructors accept the
    g256 preset of SURVEY.md §8(d) (VAE patch 16 -> 16x16x16 latents, 64 DiT tokens per frame): same kernels, other shapes."""
    vkw = dict(latent_dim=16, input_height=256, input_width=256, patch_size=16, enc_dim=128, enc_depth=1, enc_heads=2, dec_dim=128,
               dec_depth=1, dec_heads=2)
    vsd = W.synth_state_dict(W.vae_param_shapes(**vkw), seed=11)
    v = AutoencoderKL(**vkw, init_weights=False)
    v.load_state_dict(vsd)
    vcfg = O.VAEConfig(**vkw)
    g = torch.Generator().manual_seed(6)
    img = torch.rand(2, 3, 256, 256, generator=g) * 2 - 1
    with torch.no_grad():
        mom = O.vae_encode_moments(vsd, vcfg, img)
    assert rel_l2(v.encode(img).mean, mom[..., :16]) < TOL_SMALL
    dkw = dict(input_h=16, input_w=16, patch_size=2, in_channels=16, hidden_size=256, depth=1, num_heads=4, external_cond_dim=25)
    m, sd, cfg = _mk_dit(dkw, seed=12)
    x, t, a = _inputs(cfg, 2, 4, seed=13)
    with torch.no_grad():
        ref = O.dit_forward(sd, cfg, x, t, a)
    assert rel_l2(m(x, t, a), ref) < TOL_SMALL


def test_g256_geometry_full_width_forward():
    """The g256 preset at PRODUCTION widths (hidden 1024, 16 blocks, 16 heads on a 16 x 16 latent grid: 64 tokens per frame, M = 320 for the
    5-frame window and M = 64 for a context-cached step — the 64 x 96 / 64 x 48 tile shapes with 64-token frames, which the native geometry
    never runs at full width), against the CPU oracle at the north-star bound."""
    dkw = dict(input_h=16, input_w=16, patch_size=2, in_channels=16, hidden_size=1024, depth=16, num_heads=16, external_cond_dim=25)
    m, sd, cfg = _mk_dit(dkw, seed=14, max_batch=1)
    for T, actions in ((5, True), (1, False)):
        x, t, a = _inputs(cfg, 1, T, seed=15 + T, actions=actions)
        with torch.no_grad():
            ref = O.dit_forward(sd, cfg, x, t, a)
        e = rel_l2(m(x, t, a), ref)
        print(f"g256 full width T={T}: {e:.2e}")
        assert e < TOL_FULL
    m.check()


def test_constructor_variants():
    """external_cond_dim = 0 (nn.Identity in the reference, model/dit.py:263-267), a non-variational VAE
    (model/vae.py:316-317) and growing `max_frames` after construction (generate.py:139)."""
    kw = dict(SMALL_DIT, external_cond_dim=0)
    sd = W.synth_state_dict(W.dit_param_shapes(**kw), seed=21)
    assert not any(k.startswith("external_cond") for k in sd)
    m = DiT(**kw, init_weights=False, max_frames=2)
    m.load_state_dict(sd)
    cfg = O.DiTConfig(**kw)
    x, t, _ = _inputs(cfg, 1, 2, seed=3, actions=False)
    with torch.no_grad():
        ref = O.dit_forward(sd, cfg, x, t, None)
    assert rel_l2(m(x, t, None), ref) < TOL_SMALL
    m.max_frames = 6                                   # beyond the initial capacity: the handle is rebuilt transparently
    x, t, _ = _inputs(cfg, 1, 6, seed=4, actions=False)
    with torch.no_grad():
        ref = O.dit_forward(sd, cfg, x, t, None)
    assert m.max_frames == 6 and rel_l2(m(x, t, None), ref) < TOL_SMALL

    vkw = dict(SMALL_VAE)
    vsd = W.synth_state_dict(W.vae_param_shapes(**vkw, use_variational=False), seed=22)
    assert vsd["quant_conv.weight"].shape == (16, 128)
    v = AutoencoderKL(**vkw, use_variational=False, init_weights=False)
    v.load_state_dict(vsd)
    g = torch.Generator().manual_seed(8)
    img = torch.rand(2, 3, 64, 96, generator=g) * 2 - 1
    post = v.encode(img)
    # oracle: moments = quant_conv output (latent channels only); the reference pads zeros for logvar and is deterministic
    vcfg = O.VAEConfig(**vkw)
    with torch.no_grad():
        mom = O.vae_encode_moments(vsd, vcfg, img)
    assert mom.shape[-1] == 16 and rel_l2(post.mean, mom) < TOL_SMALL
    assert post.deterministic and torch.equal(post.sample(), post.mean) and post.std.abs().max().item() == 0


def test_rollout_four_prompt_frames_sliding_window():
    """generate.py's default prompt (4 frames) with max_frames = 5: the window saturates at 5 frames and slides
    (start_frame = 0, 1, 2); no actions; 6 noise steps; window and context-cached algorithms vs the oracle."""
    from gtav_amd.generate import generate_latents
    m, sd, cfg = _mk_dit(SMALL_DIT, seed=23, max_batch=1)
    g = torch.Generator().manual_seed(14)
    x0 = torch.randn(1, 4, 16, 8, 16, generator=g) * 0.5
    noise = torch.randn(1, 3, 16, 8, 16, generator=g) * 9      # exercises the +-20 clamp (generate.py:202)
    dit_fn = lambda xx, tt, aa: O.dit_forward(sd, cfg, xx, tt, aa)
    with torch.no_grad():
        ref = O.generate_latents(dit_fn, x0, 7, 6, noise, None, max_frames=5)
    out = generate_latents(m, x0, 7, 6, noise, None)
    out_c = generate_latents(m, x0, 7, 6, noise, None, ctx_cache=True)
    assert torch.equal(out[:, :4].cpu(), x0)                     # prompt frames untouched
    assert rel_l2(out, ref) < TOL_ROLLOUT and rel_l2(out_c, out) < 1e-4


# ------------------------------------------------------------------------------------------------------------------------
# round 2: harness legs (a19 / a21), direct comparisons with the reference-generated fixtures, production large-M shapes,
# fp16 saturation, stale-state guards
# ------------------------------------------------------------------------------------------------------------------------
import os  # noqa: E402

from safetensors.torch import load_file  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL_DIT_NATIVE = dict(input_h=18, input_w=32, patch_size=2, in_channels=16, hidden_size=256, depth=2, num_heads=4, external_cond_dim=25)


@pytest.fixture(scope="module")
def full_vae():
    v = VAE_models["vit-l-20-shallow-encoder"](init_weights=False, max_frames_per_call=8)
    sd = W.synth_state_dict(W.vae_param_shapes(), seed=1)
    v.load_state_dict(sd)
    return v, sd, O.vit_l_20_shallow_encoder()


def test_full_dit_vs_reference_golden_directly(full_dit):
    """HIP output against tests/golden/g3_full_dit.safetensors — outputs of the ACTUAL reference (tools/make_golden.py) — with no
    oracle in between: the headline shape (B=1, T=5, actions) and (B=2, T=3, no actions)."""
    m, _, _ = full_dit
    g = load_file(os.path.join(GOLD, "g3_full_dit.safetensors"))
    e1 = rel_l2(m(g["x_b1t5"], g["t_b1t5"], g["a_b1t5"]), g["out_b1t5"])
    e2 = rel_l2(m(g["x_b2t3"], g["t_b2t3"], None), g["out_b2t3"])
    print("full DiT vs reference golden: B=1,T=5", e1, " B=2,T=3", e2)
    assert e1 < 1e-3 and e2 < 1e-3


def test_full_vae_vs_reference_golden_directly(full_vae):
    v, _, _ = full_vae
    g = load_file(os.path.join(GOLD, "g3_full_vae.safetensors"))
    gen = torch.Generator().manual_seed(3)
    img = torch.rand(2, 3, 360, 640, generator=gen) * 2 - 1           # the fixture's input, regenerated from its seed
    post = v.encode(img)
    e_mean, e_lv = rel_l2(post.mean, g["mean"]), rel_l2(post.logvar, g["logvar"])
    dec = v.decode(g["z"])
    e_d4, e_row = rel_l2(dec[:, :, ::4, ::4], g["decoded_stride4"]), rel_l2(dec[:, :, 100], g["decoded_row100"])
    print("full VAE vs reference golden: mean", e_mean, "logvar", e_lv, "decoded stride4", e_d4, "row100", e_row)
    assert e_mean < 1e-3 and e_lv < 1e-3 and e_d4 < 1e-3 and e_row < 1e-3


def test_harness_config1_vs_reference_golden(full_vae):
    """BASELINE config 1 through the product's harness legs (a19): dummy-ramp first frame (gtav_amd.dummy_dataset) ->
    generate.vae_encode (gtav_moments_to_latents) -> 4-frame / 10-step rollout -> generate.vae_decode_frames
    (gtav_latents_to_tokens, decode, gtav_frames_to_u8), against the reference's own harness functions (fixture G7:
    generate.vae_encode, train_dit.denoise_step loop of generate.py:186-220, decode tail generate.py:238-244).
    Byte rule: the conversion kernel is bit-exact on ITS float input (layout (N,H,W,3), x255, clamp, truncation); against the
    reference bytes a pixel may differ only by +-1 and only where the reference's float value x 255 lies within the measured
    float error of an integer (the fp16 decode error moves it across the truncation boundary); the count is printed."""
    from gtav_amd.dummy_dataset import ImageDataset
    from gtav_amd.generate import generate_latents, vae_decode_frames, vae_encode
    v, _, _ = full_vae
    g = load_file(os.path.join(GOLD, "g7_harness.safetensors"))
    m, _, _ = _mk_dit(SMALL_DIT_NATIVE, seed=31, max_batch=1)
    clip = ImageDataset(split="test")[0]["video"]
    assert torch.equal(clip[None, :1, :, ::8, ::8], g["prompt_frames"])
    x0 = vae_encode(clip[None, :1].to(dev()), v, 1)
    assert x0.shape == (1, 1, 16, 18, 32)
    e_enc = rel_l2(x0, g["latents_prompt"])
    lat = generate_latents(m, x0, 4, 10, g["noise"], g["actions"])
    e_lat = rel_l2(lat, g["latents_final"])
    f32 = vae_decode_frames(lat, v, to_uint8=False)                    # (1, 4, 3, H, W)
    u8 = vae_decode_frames(lat, v, to_uint8=True)                      # (1, 4, H, W, 3)
    assert u8.dtype == torch.uint8 and u8.shape == (1, 4, 360, 640, 3)
    # (1) the byte conversion itself: exact on its own input, every pixel of every frame
    want = torch.clamp(f32.cpu().permute(0, 1, 3, 4, 2) * 255, 0, 255).byte()
    assert torch.equal(u8.cpu(), want)
    # (2) float frames vs the reference
    ref_f = g["frames_f32_stride4"]                                    # (1, 4, 90, 160, 3)
    got_f = f32.cpu().permute(0, 1, 3, 4, 2)[:, :, ::4, ::4]
    e_img = rel_l2(got_f, ref_f)
    dmax = (got_f - ref_f).abs().max().item() * 255
    # (3) bytes vs the reference bytes
    ref_u, got_u = g["frames_u8_stride4"].int(), u8.cpu()[:, :, ::4, ::4].int()
    diff = (got_u - ref_u).abs()
    near = ((ref_f * 255 - torch.round(ref_f * 255)).abs() <= dmax + 1e-3) & (ref_f * 255 > -dmax) & (ref_f * 255 < 255 + dmax)
    nbad = int((diff > 0).sum())
    print(f"harness: encode {e_enc:.2e} latents {e_lat:.2e} frames {e_img:.2e} (max |err| {dmax:.3f} of 255); "
          f"{nbad} of {diff.numel()} bytes differ from the reference, all by 1")
    assert e_enc < TOL_FULL and e_lat < TOL_SMALL and e_img < TOL_SMALL
    assert diff.max().item() <= 1 and bool(near[diff > 0].all()) and nbad < 0.08 * diff.numel()   # measured 4.2 % (7 180 of 172 800)
    assert (u8.cpu()[:, :, 100].int() - g["frames_u8_row100"].int()).abs().max().item() <= 1     # a full-width row (every column phase)


def test_encode_frames_training_leg(full_vae):
    """train.encode_frames (train_dit.py:329-351) == generate.vae_encode over all frames of a clip: vs oracle.vae_encode_frames."""
    from gtav_amd.train import encode_frames
    v, sd, cfg = full_vae
    gen = torch.Generator().manual_seed(12)
    frames = torch.rand(1, 2, 3, 360, 640, generator=gen)
    with torch.no_grad():
        ref = O.vae_encode_frames(sd, cfg, frames)
    out = encode_frames(v, frames.to(dev()))
    assert out.shape == ref.shape == (1, 2, 16, 18, 32)
    assert rel_l2(out, ref) < 1e-3


@pytest.mark.parametrize("per_call", [40, 80])
def test_encode_frames_at_the_trainer_batch(full_vae, per_call):
    """BASELINE configs[4]'s encode leg at ITS size (train_dit.py:329-351,570: 16 clips x 5 frames = 80 frames of 360x640 per step): one call of 80 frames
    (M = 46 080 tokens: 256 x 256 in-place residual tiles, 3 840-block flash attention grids, q in the exponent's unit) and two of 40 (M = 23 040: the
    persistent 128 x 192 kernel).  Frames never interact in the VAE (model/vae.py:306-322), so the oracle runs on three of them — first, last, one across the
    40-frame seam — and every frame must also agree with its own 2-frame call.  (Measured 3.4e-4 between call sizes, whatever block shapes are forced: other
    split-K slice counts change fp32 summation order, a few fp16 roundings of the stored activations flip, and six layers of softmax carry that on; both
    sizes sit 5.5e-4 from the fp32 oracle.)"""
    from gtav_amd.train import encode_frames
    _, sd, cfg = full_vae
    v = VAE_models["vit-l-20-shallow-encoder"](init_weights=False, max_frames_per_call=per_call)
    v.load_state_dict(sd)
    gen = torch.Generator().manual_seed(33)
    frames = torch.rand(16, 5, 3, 360, 640, generator=gen)
    fd = frames.to(dev())
    out = encode_frames(v, fd).cpu()                                   # (16, 5, 16, 18, 32)
    v.check()
    assert out.shape == (16, 5, 16, 18, 32) and torch.isfinite(out).all()
    for (b, t) in ((0, 0), (8, 0), (15, 4)):                           # frames 0, 40, 79
        with torch.no_grad():
            ref = O.vae_encode_frames(sd, cfg, frames[b:b + 1, t:t + 1])
        assert rel_l2(out[b:b + 1, t:t + 1], ref) < TOL_FULL
    small = VAE_models["vit-l-20-shallow-encoder"](init_weights=False, max_frames_per_call=2)
    small.load_state_dict(sd)
    for b in (3, 7, 12):
        two = encode_frames(small, fd[b:b + 1, 1:3]).cpu()
        assert rel_l2(out[b:b + 1, 1:3], two) < 6e-4


def test_full_dit_batch8_production_shapes():
    """BASELINE configs[2] forward (B=8, T=5, actions: M = 5760 tokens) at FULL size: the launch heuristic picks the large-M
    kernels here (128x192 / persistent tiles, two-slice fc2, large-M LayerNorm, 640-item attention grids) — checked as
    selected, against the CPU oracle."""
    m = DiT_models["DiT-S/2"](init_weights=False, max_batch=8)
    sd = W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0)
    m.load_state_dict(sd)
    cfg = O.dit_s_2()
    x, t, a = _inputs(cfg, 8, 5, seed=41)
    t[:, :4] = 15
    with torch.no_grad():
        ref = O.dit_forward(sd, cfg, x, t, a)
    e = rel_l2(m(x, t, a), ref)
    print("full DiT B=8 T=5 (M=5760) rel-L2", e)
    assert e < 1e-3
    m.check()
    # the context-cached sampler step of the same batch (M = 8 x 144 = 1152 tokens: the 128 x 144 frame tiles of round 4) reproduces the
    # window step on the frame being denoised; generate.py:200-220 is the loop both run
    from gtav_amd.utils import alphas_cumprod
    m.set_schedule(alphas_cumprod(1e-4))
    ad = a.to(dev())
    xd = x.to(dev()).contiguous()
    m.denoise_step_(xd, 0, 4, 15, 500, 490, False, ad)
    xc = x.to(dev()).contiguous()
    m.denoise_step_(xc, 0, 4, 15, 500, 490, False, ad)
    xc[:, -1] = x[:, -1].to(dev())
    m.denoise_step_(xc, 0, 4, 15, 500, 490, False, ad, cached=True)
    ec = rel_l2(xc[:, -1], xd[:, -1])
    print("full DiT B=8 cached step (M=1152) vs window step rel-L2", ec)
    assert ec < 1e-4      # (same kernels; the K slices of the residual GEMMs differ between the two token counts: fp32 summation order, measured 1.4e-5)
    m.check()


def test_full_dit_batch16_train_forward_loss():
    """BASELINE configs[4] DiT leg at FULL size (B=16, T=5: M = 11 520 tokens — 256x256 tiles for the N = 1024 GEMMs): v_pred,
    v_target and the loss of train.forward_loss vs oracle.train_forward_loss with injected draws."""
    from gtav_amd.train import forward_loss
    B = 16
    m = DiT_models["DiT-S/2"](init_weights=False, max_batch=B)
    sd = W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0)
    m.load_state_dict(sd)
    cfg = O.dit_s_2()
    g = torch.Generator().manual_seed(44)
    lat = torch.randn(B, 5, 16, 18, 32, generator=g) * 0.5
    a = torch.zeros(B, 5, 25)
    a[torch.arange(B)[:, None], torch.arange(5)[None], torch.randint(0, 25, (B, 5), generator=g)] = 1
    tgt = torch.randint(1, 51, (B,), generator=g)
    ctx = torch.randint(1, 41, (B,), generator=g)
    ctx_noise = torch.randn(B, 4, 16, 18, 32, generator=g)
    noise = torch.randn(B, 1, 16, 18, 32, generator=g)
    dit_fn = lambda xx, tt, aa: O.dit_forward(sd, cfg, xx, tt, aa)
    with torch.no_grad():
        loss_r, vp_r, vt_r, _, _ = O.train_forward_loss(dit_fn, lat, a, tgt, ctx, ctx_noise, noise)
    loss, vp, vt = forward_loss(m, lat, a, tgt, ctx, ctx_noise, noise)
    e = rel_l2(vp, vp_r)
    print("full DiT B=16 T=5 (M=11520) v_pred rel-L2", e, "loss", loss.item(), loss_r.item())
    assert rel_l2(vt, vt_r) < 1e-6 and e < 1e-3
    assert abs(loss.item() - loss_r.item()) / loss_r.item() < 1e-3


def test_fp16_saturation_is_clamped_and_reported():
    """A checkpoint with outlier channels: fc1 of block 0 scaled so that the GELU output exceeds the fp16 range.  The reference
    (bf16 autocast, fp32 range) would carry on; here every fp16 store saturates at +-65504 — the output stays finite — and
    gtav_dit_check reports it instead of silently producing inf/NaN."""
    from gtav_amd.lib import GtavError
    kw = dict(SMALL_DIT)
    sd = W.synth_state_dict(W.dit_param_shapes(**kw), seed=3)
    m = DiT(**kw, init_weights=False, max_batch=1)
    m.load_state_dict(sd)
    x, t, a = _inputs(O.DiTConfig(**kw), 1, 2, seed=11)
    out = m(x, t, a)
    m.check()                                            # clean run: nothing to report
    big = dict(sd)
    big["blocks.0.s_mlp.fc1.weight"] = sd["blocks.0.s_mlp.fc1.weight"] * 3e5
    m.load_state_dict(big)
    out = m(x, t, a)
    assert torch.isfinite(out).all()
    with pytest.raises(GtavError, match="fp16 range"):
        m.check()
    m.check()                                            # the error word was cleared by the failed check
    bad = x.clone()
    bad[0, 0, 0, 0, 0] = float("nan")
    m.load_state_dict(sd)
    m(bad, t, a)
    with pytest.raises(GtavError, match="NaN or inf"):
        m.check()


def test_stale_state_guards():
    """(1) a plain forward between prepare_frame_ and a cond_step >= 0 step overwrites the conditioning table -> the step
    must fail, not use corrupted adaLN rows; (2) a context-cached step without a full-window step on the same window/buffer, or
    after a forward (which rewrites the K/V caches), must fail."""
    from gtav_amd.lib import GtavError
    from gtav_amd.utils import alphas_cumprod
    m, sd, cfg = _mk_dit(SMALL_DIT, seed=4)
    g = torch.Generator().manual_seed(2)
    B, n = 2, 4
    x = torch.randn(B, n, 16, 8, 16, generator=g).to(dev()).contiguous()
    a = torch.zeros(B, n, 25, device=dev())
    m.set_schedule(alphas_cumprod(1e-4))
    m.prepare_frame_(B, n, 0, n - 1, 15, [999, 500, 0], a)
    m.denoise_step_(x, 0, n - 1, 15, 999, 500, False, a, cond_step=0)
    m.denoise_step_(x, 0, n - 1, 15, 500, 0, False, a, cached=True, cond_step=1)          # legal: cached after the window step
    xw, tw, aw = _inputs(cfg, B, 2, seed=5)
    m(xw, tw, aw)                                                                          # clobbers table + caches
    with pytest.raises(GtavError, match="prepare_frame"):
        m.denoise_step_(x, 0, n - 1, 15, 0, 0, True, a, cond_step=2)
    with pytest.raises(GtavError, match="stale"):
        m.denoise_step_(x, 0, n - 1, 15, 0, 0, True, a, cached=True)
    m.denoise_step_(x, 0, n - 1, 15, 500, 0, False, a)                                     # full window, inline conditioning: fine
    x2 = x.clone()
    with pytest.raises(GtavError, match="stale"):
        m.denoise_step_(x2, 0, n - 1, 15, 0, 0, True, a, cached=True)                      # another latent buffer
    with pytest.raises(GtavError, match="stale"):
        m.denoise_step_(x, 1, n - 1, 15, 0, 0, True, a, cached=True)                       # another window


SMALL_VAE_8x12 = dict(SMALL_VAE)                              # 64 x 96 frames, patch 8 -> 8 x 12 latents
SMALL_DIT_8x12 = dict(input_h=8, input_w=12, patch_size=2, in_channels=16, hidden_size=256, depth=2, num_heads=4, external_cond_dim=25)


def test_trainer_predict_and_predict_noise():
    """§8(f)4 inference helpers of the trainer (train_dit.py:352-552): `predict` (prompt -> autoregressive rollout with the trainer's
    constants: clamp_min 1e-6, stabilisation level 19, "W" padding of the actions -> uint8 video) and `predict_noise` (context noised
    at level - 1, last frame denoised) against the oracle with injected draws."""
    from gtav_amd.train import predict, predict_noise, stabilization_level
    assert stabilization_level(50) == 19
    vsd = W.synth_state_dict(W.vae_param_shapes(**SMALL_VAE_8x12), seed=5)
    v = AutoencoderKL(**SMALL_VAE_8x12, init_weights=False)
    v.load_state_dict(vsd)
    vcfg = O.VAEConfig(**SMALL_VAE_8x12)
    m, sd, cfg = _mk_dit(SMALL_DIT_8x12, seed=17, max_batch=1)
    dit_fn = lambda xx, tt, aa: O.dit_forward(sd, cfg, xx, tt, aa)
    g = torch.Generator().manual_seed(21)
    frames = torch.rand(2, 5, 3, 64, 96, generator=g)
    acts = torch.zeros(2, 5, 25)
    acts[:, -1, 1] = 1
    nz = torch.randn(1, 3, 16, 8, 12, generator=g) * 6
    with torch.no_grad():
        ref = O.trainer_predict_latents(dit_fn, vsd, vcfg, frames, acts, nz, num_frames=7, n_prompt_frames=4, ddim_noise_steps_inference=5)
        ref_u8 = O.vae_decode_latents(vsd, vcfg, ref)
    lat, video = predict(m, v, frames, acts, nz, num_frames=7, n_prompt_frames=4, ddim_noise_steps_inference=5)
    assert video.shape == (1, 7, 64, 96, 3) and video.dtype == torch.uint8
    assert rel_l2(lat, ref) < TOL_ROLLOUT
    assert (video.cpu().int() - ref_u8.int()).abs().max().item() <= 1
    cn = torch.randn(1, 4, 16, 8, 12, generator=g) * 6
    nf = torch.randn(1, 1, 16, 8, 12, generator=g)
    with torch.no_grad():
        l_r, old_r, new_r = O.trainer_predict_noise(dit_fn, vsd, vcfg, frames, acts, cn, nf, ddim_noise_steps_inference=5)
    l_g, old_g, new_g = predict_noise(m, v, frames, acts, cn, nf, ddim_noise_steps_inference=5)
    assert rel_l2(l_g, l_r) < TOL_SMALL and rel_l2(old_g, old_r) < TOL_SMALL and rel_l2(new_g, new_r) < TOL_ROLLOUT


def test_dataset_step_and_prompt_resize():
    """§8(f)3 / (f)2: the per-sample dataset transform (ToTensor + SplitImages + Resize((360, 640)) of a 270 x 2400 strip) and the
    prompt-frame resize (arbitrary size -> 360 x 640, incl. down-scaling with antialiasing) as HIP kernels vs the oracle
    (torch's antialiased bilinear interpolate, which torchvision's Resize calls)."""
    from gtav_amd.data import resize_frames, strip_to_clip
    g = torch.Generator().manual_seed(5)
    strip = torch.randint(0, 256, (270, 2400, 3), generator=g, dtype=torch.uint8)
    clip = strip_to_clip(strip)
    ref = O.strip_to_clip(strip)
    assert clip.shape == ref.shape == (5, 3, 360, 640)
    assert (clip.cpu() - ref).abs().max().item() < 2e-6
    for (H, W_) in ((720, 1280), (500, 333), (360, 640), (123, 77)):
        x = torch.rand(2, 3, H, W_, generator=g)
        got = resize_frames(x.to(dev()))
        want = O.resize_frames(x)
        assert got.shape == want.shape == (2, 3, 360, 640)
        assert (got.cpu() - want).abs().max().item() < 2e-6, (H, W_)


def test_prompt_resize_against_the_committed_fixture():
    """§8(f)2 pinned: the HIP resize kernel (data.resize_frames) against fixture G11 — committed output rows of `transforms.Resize((360, 640))` as the reference
    applies it (generate.py:150-153, web_dataset.py:105-107), generated by tools/make_golden.py — on the 500 x 333 probe (down- and up-scaling at once),
    the dataset's 480 x 270 frames and a 1280 x 720 image; no oracle call at run time."""
    from safetensors.torch import load_file
    from helpers import G11_ROWS, G11_SIZES, resize_probe_image
    from gtav_amd.data import resize_frames
    g = load_file(os.path.join(os.path.dirname(__file__), "golden", "g11_resize.safetensors"))
    for tag, (H, W_) in G11_SIZES.items():
        img = resize_probe_image(H, W_).permute(2, 0, 1)[None].float() / 255.0
        y = resize_frames(img.to(dev())).cpu()[0]
        assert y.shape == (3, 360, 640)
        assert (y[:, list(G11_ROWS)] - g[f"{tag}.rows"]).abs().max().item() < 2e-6, tag
        assert (y[:, ::8, ::8] - g[f"{tag}.stride8"]).abs().max().item() < 2e-6, tag


def test_read_prompt_frame_and_video_out(tmp_path):
    """generate.py:150-153 + 244-246 around the path: a PNG start frame -> prompt tensor; generated uint8 frames -> a video file."""
    from PIL import Image
    from gtav_amd.data import read_avi_mjpeg, read_prompt_frame, write_video
    g = torch.Generator().manual_seed(6)
    img = torch.randint(0, 256, (300, 500, 3), generator=g, dtype=torch.uint8)
    p = str(tmp_path / "start.png")
    Image.fromarray(img.numpy(), "RGB").save(p)
    x = read_prompt_frame(p)
    assert x.shape == (1, 1, 3, 360, 640)
    want = O.resize_frames(img.permute(2, 0, 1)[None].float() / 255.0)
    assert (x[0].cpu() - want).abs().max().item() < 2e-6
    # a smooth clip (JPEG is lossy: noise would not survive it) through the uint8 tail and the video writer
    from gtav_amd.dummy_dataset import ImageDataset
    frames = (ImageDataset(split="test")[0]["video"][:3].permute(0, 2, 3, 1) * 255).clamp(0, 255).byte()
    out = write_video(str(tmp_path / "clip.mp4"), frames, fps=10)          # no torchvision here: lands as Motion-JPEG AVI
    back = read_avi_mjpeg(out)
    assert back.shape == frames.shape and (back.int() - frames.int()).abs().float().mean().item() < 2.0


def test_config0_production_size_vs_reference_golden(full_dit, full_vae):
    """BASELINE configs[0] at PRODUCTION size (fixture G9: the reference's DiT-S/2 — 16 blocks, 608 M parameters — and its full ViT-L/20 VAE
    through the reference's own vae_encode / denoise_step loop / decode tail, generate.py:50-66,186-244): dummy-ramp prompt frame ->
    4-frame / 10-step rollout (33 full-size forwards, windows of 2, 3 and 4 frames) -> frames and bytes, through the product's harness."""
    from gtav_amd.dummy_dataset import ImageDataset
    from gtav_amd.generate import generate_latents, vae_decode_frames, vae_encode
    m, _, _ = full_dit
    v, _, _ = full_vae
    g = load_file(os.path.join(GOLD, "g9_config0_full.safetensors"))
    clip = ImageDataset(split="test")[0]["video"]
    assert torch.equal(clip[None, :1, :, ::8, ::8], g["prompt_frames"])
    x0 = vae_encode(clip[None, :1].to(dev()), v, 1)
    e_enc = rel_l2(x0, g["latents_prompt"])
    lat = generate_latents(m, x0, 4, 10, g["noise"], g["actions"])
    e_lat = rel_l2(lat, g["latents_final"])
    lat_c = generate_latents(m, x0, 4, 10, g["noise"], g["actions"], ctx_cache=True)
    f32 = vae_decode_frames(lat, v, to_uint8=False)
    u8 = vae_decode_frames(lat, v, to_uint8=True)
    ref_f = g["frames_f32_stride4"]
    got_f = f32.cpu().permute(0, 1, 3, 4, 2)[:, :, ::4, ::4]
    e_img = rel_l2(got_f, ref_f)
    dmax = (got_f - ref_f).abs().max().item() * 255
    diff = (u8.cpu()[:, :, ::4, ::4].int() - g["frames_u8_stride4"].int()).abs()
    print(f"config0 full size: encode {e_enc:.2e} latents {e_lat:.2e} (cached vs window {rel_l2(lat_c, lat):.1e}) frames {e_img:.2e} "
          f"(max |err| {dmax:.3f} of 255); {int((diff > 0).sum())} of {diff.numel()} bytes differ, max {int(diff.max())}")
    # 33 chained full-size forwards: per-forward error 6-9e-4 (TOL_FULL), accumulated over the rollout
    # (measured: latents 3.7e-4, frames 4.5e-4, byte differences of at most 1 — profiles/round3/test_margins_pytest_s.txt)
    assert e_enc < TOL_FULL and e_lat < 1.5e-3 and e_img < 1.5e-3 and rel_l2(lat_c, lat) < 1e-4
    assert diff.max().item() <= 1 and int((diff > 0).sum()) < 0.08 * diff.numel()   # measured 4.4 % (7 622 of 172 800)


def test_full_size_checkpoint_in_the_reference_writers_convention(tmp_path, full_dit):
    """a FULL-SIZE checkpoint file as the reference's own writer leaves it — `accelerator.save(unwrap(dit).state_dict(),
    path, safe_serialization=True)` (train_dit.py:758-762) de-duplicates the 34 shared rotary `freqs` aliases down to `spatial_rotary_emb.freqs`
    / `temporal_rotary_emb.freqs` — loaded through load_state_dict_file -> load_state_dict -> the C-ABI on the GPU, gives the
    same forward as the in-memory weights; and train.save_model writes that same convention back (key set and values equal)."""
    from safetensors.torch import save_file
    from gtav_amd.train import save_model
    from gtav_amd.weights import load_state_dict_file
    m, sd, cfg = full_dit
    x, t, a = _inputs(cfg, 1, 3, seed=23)
    want = m(x, t, a).clone()
    path = str(tmp_path / "dit_epoch_1_920000.safetensors")
    on_disk = {k: v.contiguous() for k, v in sd.items()}
    on_disk["spatial_rotary_emb.freqs"] = W.rope_freqs_pixel(32, 256)
    on_disk["temporal_rotary_emb.freqs"] = W.rope_freqs_lang(64)
    save_file(on_disk, path)                                           # 2.4 GB of fp32, 324 entries
    m2 = DiT_models["DiT-S/2"](init_weights=False, max_batch=1)
    missing, unexpected = m2.load_state_dict(load_state_dict_file(path))
    assert not missing and not unexpected
    assert torch.equal(m2(x, t, a), want)
    out = str(tmp_path / "resaved.safetensors")
    save_model(m2, out)
    back = load_state_dict_file(out)
    assert set(back) == set(on_disk)
    for k in ("blocks.7.t_attn.to_qkv.weight", "final_layer.adaLN_modulation.1.bias", "spatial_rotary_emb.freqs", "temporal_rotary_emb.freqs"):
        assert torch.equal(back[k], on_disk[k]), k


def test_visualize_step_debug_grid(tmp_path):
    """utils.visualize_step (reference utils.py:104-211): same signature; the three image rows are the VAE decode of the original / noisy /
    denoised latents clamped to [0, 1] (against the oracle decode), x_start = (x_noisy - sqrt(1 - a) v) / sqrt(a) when no `pred` is given, and a
    PNG lands where the reference puts it."""
    from gtav_amd.utils import alphas_cumprod, visualize_step
    vsd = W.synth_state_dict(W.vae_param_shapes(**SMALL_VAE), seed=5)
    v = AutoencoderKL(**SMALL_VAE, init_weights=False)
    v.load_state_dict(vsd)
    vcfg = O.VAEConfig(**SMALL_VAE)
    g = torch.Generator().manual_seed(4)
    h, w = 64 // 8, 96 // 8
    x = torch.randn(1, 3, 16, h, w, generator=g) * 0.05
    nz = torch.randn(1, 3, 16, h, w, generator=g)
    ac = alphas_cumprod(1e-6)
    step = 400
    xn = ac[step].sqrt() * x + (1 - ac[step]).sqrt() * nz
    vp = torch.randn(1, 3, 16, h, w, generator=g) * 0.1
    path, imgs = visualize_step(x, xn, nz, vp, step, v, ac, name="grid.png", out_dir=str(tmp_path))
    assert os.path.getsize(path) > 10_000
    with torch.no_grad():
        ref = lambda lat: ((O.vae_decode(vsd, vcfg, (lat / 0.07843137255).reshape(3, 16, h * w).permute(0, 2, 1)) + 1) / 2).clamp(0, 1)
        assert rel_l2(imgs["orig"][0], ref(x[0])) < TOL_SMALL and rel_l2(imgs["noisy"][0], ref(xn[0])) < TOL_SMALL
        xs = (xn - (1 - ac[step]).sqrt() * vp) / ac[step].sqrt()
        assert rel_l2(imgs["denoised"][0], ref(xs[0])) < TOL_SMALL
    assert float(imgs["orig"].min()) >= 0.0 and float(imgs["orig"].max()) <= 1.0
