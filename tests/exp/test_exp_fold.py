"""LayerNorm fold (gemm.h EPI_*_FOLD): the LayerNorm + adaLN modulate of model/dit.py:19-27,200-225 executed inside the
epilogues of the residual GEMM in front of it and of the GEMM behind it.  Parity of the folded path against the CPU oracle (same bound
as the unfolded path: 1e-3 relative L2 per forward at full size) and against the unfolded path, on every block shape the launch
heuristic picks for the folded epilogues (64 x 48 / 64 x 96 / 128 x 96 loader / 128 x 128 / 128 x 192 tiles), through the plain
forward, the prepared sampler step (per-frame tables + gathered current-step rows), the captured graph and the context-cached step.

EXPERIMENTS BUILD ONLY (the fold measured slower at every size and left the product library in round 5; its kernels stay in libgtav_amd_exp.so for tools/):
run with  GTAV_TEST_EXP=1 python -m pytest tests/exp -q  — the session then loads libgtav_amd_exp.so instead of the product library."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = [pytest.mark.gpu, pytest.mark.exp]   # experiments build only: GTAV_TEST_EXP=1 python -m pytest tests/exp -q (tests/conftest.py)

from helpers import dev  # noqa: E402
from helpers import rel_l2  # noqa: E402
from oracle import ref_cpu as O  # noqa: E402
import gtav_amd.weights as W  # noqa: E402
from gtav_amd.model.dit import DiT, DiT_models  # noqa: E402

G64 = dict(input_h=16, input_w=16, patch_size=2, in_channels=16, hidden_size=256, depth=2, num_heads=4, external_cond_dim=25)   # 64 tokens per frame


def _mk(kw, seed, max_batch):
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


@pytest.mark.parametrize("B,T,actions", [(2, 4, True), (1, 5, False), (6, 5, True), (3, 1, True)])
def test_fold_small_model_matches_oracle_and_unfolded(B, T, actions):
    """hidden 256, two blocks, 64 tokens per frame: 64-1920 tokens (skinny / 64 x 96 / loader tiles; frames shorter than a block tile, so
    a tile spans up to four frames), with and without actions; T = 1 is the shape of a context-cached step."""
    m, sd, cfg = _mk(G64, seed=21, max_batch=B)
    x, t, a = _inputs(cfg, B, T, seed=5 + B, actions=actions)
    with torch.no_grad():
        ref = O.dit_forward(sd, cfg, x, t, a)
    m.set_fold(0)
    plain = m(x, t, a).clone()
    m.set_fold(2)
    fold = m(x, t, a).clone()
    again = m(x, t, a)
    e0, e1, d = rel_l2(plain, ref), rel_l2(fold, ref), rel_l2(fold, plain)
    print(f"G64 B={B} T={T}: unfolded {e0:.2e}  folded {e1:.2e}  folded vs unfolded {d:.2e}")
    assert e0 < 2e-3 and e1 < 2e-3 and d < 2e-3
    assert torch.equal(fold, again)          # fixed summation order: run-to-run bit-identical
    m.check()


@pytest.fixture(scope="module")
def full_dit():
    m = DiT_models["DiT-S/2"](init_weights=False, max_batch=2)
    sd = W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0)
    m.load_state_dict(sd)
    return m, sd, O.dit_s_2()


@pytest.mark.parametrize("B,T", [(1, 5), (2, 3), (1, 1)])
def test_fold_full_size_every_seam(full_dit, B, T):
    """DiT-S/2 at native geometry (144 tokens per frame) with every seam folded (mode 2): M = 720 / 864 / 144."""
    m, sd, cfg = full_dit
    x, t, a = _inputs(cfg, B, T, seed=50 + B)
    with torch.no_grad():
        ref = O.dit_forward(sd, cfg, x, t, a)
    try:
        m.set_fold(0)
        plain = m(x, t, a).clone()
        m.set_fold(2)
        fold = m(x, t, a).clone()
        assert torch.equal(fold, m(x, t, a))
    finally:
        m.set_fold(1, -1, -1)
    e0, e1, d = rel_l2(plain, ref), rel_l2(fold, ref), rel_l2(fold, plain)
    print(f"full DiT B={B} T={T}: unfolded {e0:.2e}  folded {e1:.2e}  folded vs unfolded {d:.2e}")
    assert e0 < 1e-3 and e1 < 1e-3 and d < 1e-3
    m.check()


def test_fold_seam_a_only_and_thresholds(full_dit):
    """mode 1 with thresholds: seam A (out-proj -> fc1) folded, seam B not (the fc2 slabs are reduced by the LayerNorm kernel as before), and
    the other way round."""
    m, sd, cfg = full_dit
    x, t, a = _inputs(cfg, 1, 5, seed=77)
    with torch.no_grad():
        ref = O.dit_forward(sd, cfg, x, t, a)
    try:
        for ta, tb in ((0, 1 << 30), (1 << 30, 0)):
            m.set_fold(1, ta, tb)
            e = rel_l2(m(x, t, a), ref)
            print(f"thresholds a={ta} b={tb}: {e:.2e}")
            assert e < 1e-3
    finally:
        m.set_fold(1, 1 << 30, 1 << 30)


def test_fold_batch8_production_shapes():
    """BASELINE configs[2] forward (B = 8, T = 5: M = 5760) with both seams folded (128 x 192 tiles, full-K fc2) against the oracle and against
    the unfolded path (the default policy: the fold measured slower on MI355X)."""
    m = DiT_models["DiT-S/2"](init_weights=False, max_batch=8)
    sd = W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0)
    m.load_state_dict(sd)
    cfg = O.dit_s_2()
    x, t, a = _inputs(cfg, 8, 5, seed=41)
    t[:, :4] = 15
    with torch.no_grad():
        ref = O.dit_forward(sd, cfg, x, t, a)
    plain = m(x, t, a).clone()                    # default policy
    m.set_fold(0)
    assert torch.equal(m(x, t, a), plain)         # ... which is the unfolded path
    m.set_fold(2)
    fold = m(x, t, a).clone()
    e0, e1, d = rel_l2(plain, ref), rel_l2(fold, ref), rel_l2(fold, plain)
    print(f"full DiT B=8 T=5 (M=5760): unfolded {e0:.2e}  folded {e1:.2e}  folded vs unfolded {d:.2e}")
    assert e0 < 1e-3 and e1 < 1e-3 and d < 1e-3
    m.check()


def test_fold_sampler_prepared_steps_graph_and_cached():
    """generate_latents on the 64-token geometry with every seam folded: per-frame tables (prepare_frame), gathered current-step rows,
    eager warm-up -> capture -> replay, window vs context-cached steps; against the unfolded rollout and the oracle rollout."""
    from gtav_amd.generate import generate_latents
    m, sd, cfg = _mk(G64, seed=33, max_batch=2)
    g = torch.Generator().manual_seed(9)
    B, n_prompt, total, steps = 2, 2, 5, 4
    x0 = torch.randn(B, n_prompt, 16, 16, 16, generator=g) * 0.5
    nz = torch.randn(B, total - n_prompt, 16, 16, 16, generator=g)
    a = torch.zeros(B, total, 25)
    a[:, :, 3] = 1
    dit_fn = lambda xx, tt, aa: O.dit_forward(sd, cfg, xx, tt, aa)
    with torch.no_grad():
        ref = O.generate_latents(dit_fn, x0, total, steps, nz, a)
    outs = {}
    for mode in (0, 2):
        m.set_fold(mode)
        outs[mode] = generate_latents(m, x0, total, steps, nz, a).cpu()
        outs[(mode, "cached")] = generate_latents(m, x0, total, steps, nz, a, ctx_cache=True).cpu()
        outs[(mode, "inline")] = generate_latents(m, x0, total, steps, nz, a, hoist_cond=False).cpu()
    for k, v in outs.items():
        e = rel_l2(v, ref)
        print(f"rollout {k}: {e:.2e}")
        assert e < 3e-3, k
    assert rel_l2(outs[2], outs[0]) < 5e-3
    assert rel_l2(outs[(2, "cached")], outs[2]) < 1e-4          # same kernels on the same rows up to the tile shapes' summation order
    assert torch.equal(outs[(2, "inline")], outs[2])            # hoisted vs per-step tables: the same arithmetic
    m.check()
