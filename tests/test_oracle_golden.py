"""CPU: pins oracle/ref_cpu.py (and the product's host-side tables) against golden vectors captured from the ACTUAL
reference (tools/make_golden.py -> tests/golden/*.safetensors).  fp32 vs fp32, so the bound is tight (2e-5 rel-L2;
most cases are bit-exact because the oracle uses the same ATen CPU kernels)."""
import os

import pytest
import torch
from safetensors.torch import load_file

from oracle import ref_cpu as O
import gtav_amd.weights as W

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL_DIT = dict(input_h=8, input_w=16, patch_size=2, in_channels=16, hidden_size=256, depth=2, num_heads=4, external_cond_dim=25)
SMALL_VAE = dict(latent_dim=16, input_height=64, input_width=96, patch_size=8, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=256,
                 dec_depth=2, dec_heads=4)
TOL = 2e-5


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def gold(name):
    return load_file(os.path.join(GOLD, name))


def test_g0_schedule_and_known_answers():
    g = gold("g0_constants.safetensors")
    for tag, cm in (("gen", 1e-4), ("train", 1e-6)):
        assert torch.equal(O.sigmoid_beta_schedule(1000, clamp_min=cm), g[f"betas64_{tag}"])
        assert torch.equal(O.alphas_cumprod_table(cm), g[f"alphas_cumprod_{tag}"])
    # known answers recorded in SURVEY.md §8(a) a20
    ac = O.alphas_cumprod_table(1e-4)
    for i, v in ((0, 0.99969977), (15, 0.99499559), (499, 0.50005001), (999, 1.0000775e-4)):
        assert abs(ac[i].item() - v) / v < 2e-6
    assert torch.equal(O.noise_range_generate(100), g["noise_range_gen100"])
    assert [int(v) for v in O.noise_range_generate(100)] == g["noise_range_gen100_long"].tolist()
    assert [int(v) for v in O.noise_range_generate(100)][:4] == [0, 9, 19, 29]
    assert torch.equal(O.noise_range_train(50), g["noise_range_train50"])


def test_g0_rope_tables_and_embeddings():
    g = gold("g0_constants.safetensors")
    assert torch.equal(O.rope_freqs_pixel(32, 256), g["rope_spatial_freqs"])
    assert torch.equal(O.rope_angles_axial(9, 16, O.rope_freqs_pixel(32, 256)), g["rope_spatial_angles_9x16"])
    assert torch.equal(O.rope_freqs_lang(64), g["rope_temporal_freqs"])
    assert torch.equal(O.rope_angles_temporal(5, O.rope_freqs_lang(64)), g["rope_temporal_angles_T5"])
    assert torch.equal(O.vae_rope_angles(O.vit_l_20_shallow_encoder(), 16, 1024), g["rope_vae_angles_18x32"])
    assert torch.equal(O.timestep_embedding(torch.tensor([0, 15, 19, 500, 999])), g["timestep_embedding_rows"])
    m = O.modulate(torch.ones(1, 1, 1, 1, 1), torch.full((1, 1, 1), 0.5), torch.full((1, 1, 1), 0.25))
    assert torch.equal(m, g["modulate_known"]) and abs(m.item() - 1.75000095) < 1e-6
    assert torch.allclose(O.dummy_clip().mean(dim=(2, 3)), g["dummy_clip_means"])
    assert torch.equal(O.actions_to_one_hot([-1, 3, 0, 24, -1]), g["one_hot_example"])


def test_product_host_tables_match_reference():
    """The tables the product uploads (computed in gtav_amd.model.dit with torch CPU ops) equal the reference's."""
    from gtav_amd.model.dit import _rope_tables_axial, _rope_tables_temporal, _timestep_table
    from gtav_amd.utils import alphas_cumprod
    from gtav_amd.dummy_dataset import ImageDataset, actions_to_one_hot
    g = gold("g0_constants.safetensors")
    c, s = _rope_tables_axial(W.rope_freqs_pixel(32, 256), 9, 16)
    ang = g["rope_spatial_angles_9x16"].reshape(144, 64)
    assert torch.equal(c, ang.cos()) and torch.equal(s, ang.sin())
    c, s = _rope_tables_temporal(W.rope_freqs_lang(64), 5)
    assert torch.equal(c, g["rope_temporal_angles_T5"].cos()) and torch.equal(s, g["rope_temporal_angles_T5"].sin())
    c, s = _rope_tables_axial(W.rope_freqs_pixel(16, 576), 18, 32)
    va = g["rope_vae_angles_18x32"].reshape(576, 32)
    assert torch.equal(c[:, :32], va.cos()) and torch.equal(s[:, :32], va.sin())
    assert torch.equal(c[:, 32:], torch.ones(576, 32)) and torch.equal(s[:, 32:], torch.zeros(576, 32))
    tab = _timestep_table()
    assert torch.equal(tab[[0, 15, 19, 500, 999]], g["timestep_embedding_rows"])
    assert torch.equal(alphas_cumprod(1e-4), g["alphas_cumprod_gen"]) and torch.equal(alphas_cumprod(1e-6), g["alphas_cumprod_train"])
    assert torch.allclose(ImageDataset("test").sequence_blue_red.mean(dim=(2, 3)), g["dummy_clip_means"])
    assert torch.equal(actions_to_one_hot([-1, 3, 0, 24, -1]), g["one_hot_example"])


def test_g2_small_dit():
    g = gold("g2_small_dit.safetensors")
    sd = W.synth_state_dict(W.dit_param_shapes(**SMALL_DIT), seed=3)
    cfg = O.DiTConfig(**SMALL_DIT)
    taps = {}
    with torch.no_grad():
        assert rel(O.dit_forward(sd, cfg, g["x"], g["t"], g["actions"], taps), g["out_actions"]) < TOL
        assert rel(O.dit_forward(sd, cfg, g["x"], g["t"], None), g["out_noactions"]) < TOL
    for i in range(2):
        assert rel(taps[f"block{i}"], g[f"block{i}_actions"]) < TOL


def test_g2_small_vae():
    g = gold("g2_small_vae.safetensors")
    sd = W.synth_state_dict(W.vae_param_shapes(**SMALL_VAE), seed=5)
    cfg = O.VAEConfig(**SMALL_VAE)
    with torch.no_grad():
        mom = O.vae_encode_moments(sd, cfg, g["img"])
        assert rel(mom[..., :16], g["mean"]) < TOL
        assert rel(mom[..., 16:].clamp(-30, 20), g["logvar"]) < TOL
        assert rel(O.vae_decode(sd, cfg, g["z"]), g["decoded"]) < TOL


def test_g4_denoise_step():
    g = gold("g4_denoise_step.safetensors")
    sd = W.synth_state_dict(W.dit_param_shapes(**SMALL_DIT), seed=4)
    cfg = O.DiTConfig(**SMALL_DIT)
    ac = O.alphas_cumprod_table(1e-4)[:, None, None, None]
    nr = torch.linspace(0, 999, 11)
    fn = lambda x, t, a: O.dit_forward(sd, cfg, x, t, a)
    with torch.no_grad():
        for idx in (10, 4, 0):
            xp, vp = O.denoise_step(fn, g["x"], g["actions"], idx, 15, nr, ac, start_frame=1)
            assert rel(xp, g[f"x_pred_{idx}"]) < TOL and rel(vp, g[f"v_pred_{idx}"]) < TOL


def test_g5_rollout():
    g = gold("g5_rollout.safetensors")
    sd = W.synth_state_dict(W.dit_param_shapes(**SMALL_DIT), seed=6)
    cfg = O.DiTConfig(**SMALL_DIT)
    fn = lambda x, t, a: O.dit_forward(sd, cfg, x, t, a)
    with torch.no_grad():
        out = O.generate_latents(fn, g["x_prompt"], 4, 10, g["noise"], g["actions"])
    assert rel(out, g["latents"]) < 1e-4     # 33 chained fp32 forwards


def test_g6_train_forward_loss():
    g = gold("g6_train_forward.safetensors")
    sd = W.synth_state_dict(W.dit_param_shapes(**SMALL_DIT), seed=7)
    cfg = O.DiTConfig(**SMALL_DIT)
    fn = lambda x, t, a: O.dit_forward(sd, cfg, x, t, a)
    with torch.no_grad():
        loss, vp, vt, xn, t = O.train_forward_loss(fn, g["latents"], g["actions"], g["target_idx"], g["ctx_idx"], g["ctx_noise"], g["noise"])
    assert torch.equal(t, g["t"])
    assert rel(xn, g["x_noisy"]) < 1e-6 and rel(vt, g["v_target"]) < 1e-6
    assert rel(vp, g["v_pred"]) < TOL and abs(loss.item() - g["loss"].item()) / g["loss"].item() < TOL


@pytest.mark.slow
def test_g3_full_dit():
    g = gold("g3_full_dit.safetensors")
    sd = W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0)
    cfg = O.dit_s_2()
    taps = {}
    with torch.no_grad():
        out = O.dit_forward(sd, cfg, g["x_b1t5"], g["t_b1t5"], g["a_b1t5"], taps)
        assert rel(out, g["out_b1t5"]) < TOL
        for i in (0, 7, 15):
            tp = taps[f"block{i}"]
            st = torch.stack([tp.mean(), tp.abs().max(), tp.abs().mean()])
            assert torch.allclose(st, g[f"block{i}_stats"], rtol=1e-4)
            assert torch.allclose(tp.reshape(-1)[:: max(1, tp.numel() // 8)][:8], g[f"block{i}_samples"], rtol=1e-4, atol=1e-4)
        assert rel(O.dit_forward(sd, cfg, g["x_b2t3"], g["t_b2t3"], None), g["out_b2t3"]) < TOL


@pytest.mark.slow
def test_g3_full_vae():
    g = gold("g3_full_vae.safetensors")
    sd = W.synth_state_dict(W.vae_param_shapes(), seed=1)
    cfg = O.vit_l_20_shallow_encoder()
    gen = torch.Generator().manual_seed(3)
    img = torch.rand(2, 3, 360, 640, generator=gen) * 2 - 1
    z = torch.randn(2, 576, 16, generator=gen)
    assert torch.equal(z, g["z"])
    with torch.no_grad():
        mom = O.vae_encode_moments(sd, cfg, img)
        assert rel(mom[..., :16], g["mean"]) < TOL
        dec = O.vae_decode(sd, cfg, z)
        assert rel(dec[:, :, ::4, ::4], g["decoded_stride4"]) < TOL and rel(dec[:, :, 100], g["decoded_row100"]) < TOL


@pytest.mark.slow
def test_g7_harness_encode_rollout_decode_bytes():
    """G7 (reference generate.vae_encode -> denoise_step rollout -> decode tail, BASELINE config 1): pins the oracle's harness
    restatements `vae_encode_frames`, `generate_latents`, `vae_decode_latents` (oracle/ref_cpu.py:424-456) and the dummy clip,
    including the uint8 tail: bytes equal to the reference except where fp32 summation-order noise (<= 2e-5 relative) moves a
    value across a truncation boundary — at most a handful of pixels, each by 1."""
    g = gold("g7_harness.safetensors")
    vsd = W.synth_state_dict(W.vae_param_shapes(), seed=1)
    vcfg = O.vit_l_20_shallow_encoder()
    kw = dict(input_h=18, input_w=32, patch_size=2, in_channels=16, hidden_size=256, depth=2, num_heads=4, external_cond_dim=25)
    sd = W.synth_state_dict(W.dit_param_shapes(**kw), seed=31)
    cfg = O.DiTConfig(**kw)
    clip = O.dummy_clip()
    assert torch.equal(clip[None, :1, :, ::8, ::8], g["prompt_frames"])
    fn = lambda x, t, a: O.dit_forward(sd, cfg, x, t, a)
    with torch.no_grad():
        x0 = O.vae_encode_frames(vsd, vcfg, clip[None, :1])
        assert rel(x0, g["latents_prompt"]) < TOL
        lat = O.generate_latents(fn, x0, 4, 10, g["noise"], g["actions"])
        assert rel(lat, g["latents_final"]) < 1e-4
        u8 = O.vae_decode_latents(vsd, vcfg, g["latents_final"])
    assert u8.dtype == torch.uint8 and u8.shape == (1, 4, 360, 640, 3)
    d = (u8[:, :, ::4, ::4].int() - g["frames_u8_stride4"].int()).abs()
    assert d.max().item() <= 1 and int((d > 0).sum()) <= 20, int((d > 0).sum())
    assert (u8[:, :, 100].int() - g["frames_u8_row100"].int()).abs().max().item() <= 1


def test_g10_frame_loop_three_target_frames():
    """G10 (reference statements of train_dit.py:590-682 on the reference DiT module, 7-frame clips = three target frames with windows of
    5 frames sliding): pins `oracle.train_shared_step` — per-frame losses, v_pred of every target frame, the returned mean loss."""
    g = gold("g10_frame_loop.safetensors")
    sd = W.synth_state_dict(W.dit_param_shapes(**SMALL_DIT), seed=17)
    cfg = O.DiTConfig(**SMALL_DIT)
    fn = lambda x, t, a: O.dit_forward(sd, cfg, x, t, a)
    with torch.no_grad():
        mean, losses, vps, vts = O.train_shared_step(fn, g["latents"], g["actions"], g["target_idx"], g["ctx_idx"],
                                                     [g[f"ctx_noise{k}"] for k in range(3)], [g[f"noise{k}"] for k in range(3)])
    for k in range(3):
        assert rel(vts[k], g[f"v_target{k}"]) < 1e-6 and rel(vps[k], g[f"v_pred{k}"]) < TOL
        assert abs(losses[k].item() - g[f"loss{k}"].item()) / g[f"loss{k}"].item() < TOL
    assert abs(mean.item() - g["mean_loss"].item()) / g["mean_loss"].item() < TOL


@pytest.mark.slow
def test_g9_config0_production_size():
    """G9 = BASELINE configs[0] at PRODUCTION size (reference DiT-S/2, 16 blocks, 608 M parameters + the full ViT-L/20 VAE through the
    reference's own vae_encode / denoise_step loop / decode tail): the oracle's 33 chained full-size forwards (windows of 2, 3, 4
    frames), stride-sampled frames and bytes."""
    g = gold("g9_config0_full.safetensors")
    vsd = W.synth_state_dict(W.vae_param_shapes(), seed=1)
    vcfg = O.vit_l_20_shallow_encoder()
    sd = W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0)
    cfg = O.dit_s_2()
    clip = O.dummy_clip()
    assert torch.equal(clip[None, :1, :, ::8, ::8], g["prompt_frames"])
    fn = lambda x, t, a: O.dit_forward(sd, cfg, x, t, a)
    with torch.no_grad():
        x0 = O.vae_encode_frames(vsd, vcfg, clip[None, :1])
        assert rel(x0, g["latents_prompt"]) < TOL
        lat = O.generate_latents(fn, x0, 4, 10, g["noise"], g["actions"])
        assert rel(lat, g["latents_final"]) < 1e-4
        u8 = O.vae_decode_latents(vsd, vcfg, g["latents_final"])
    d = (u8[:, :, ::4, ::4].int() - g["frames_u8_stride4"].int()).abs()
    assert d.max().item() <= 1 and int((d > 0).sum()) <= 20, int((d > 0).sum())
    assert (u8[:, :, 100].int() - g["frames_u8_row100"].int()).abs().max().item() <= 1


def test_initialize_weights_statistics():
    """a13: DiT.initialize_weights (model/dit.py:295-326) — N(0, 0.02) Linear weights, zero biases, t-MLP std 0.01, the adaLN
    modulation of every block ZEROED (blocks are the identity at init), final adaLN std 0.01, final linear std 0.001;
    AutoencoderKL.initialize_weights (model/vae.py:239-256) — xavier-uniform Linears, zero biases, LayerNorm (1, 0)."""
    from gtav_amd.model.dit import DiT
    from gtav_amd.model.vae import AutoencoderKL
    torch.manual_seed(0)
    m = DiT(input_h=8, input_w=16, hidden_size=256, depth=2, num_heads=4)          # init_weights=True is the default
    sd = m.state_dict()

    def std_of(k):
        return sd[k].std().item()

    for k, v in sd.items():
        if k.endswith(".bias"):
            assert v.abs().max().item() == 0, k
    for i in range(2):
        for h in "st":
            assert sd[f"blocks.{i}.{h}_adaLN_modulation.1.weight"].abs().max().item() == 0
            assert abs(std_of(f"blocks.{i}.{h}_attn.to_qkv.weight") - 0.02) < 0.001
            assert abs(std_of(f"blocks.{i}.{h}_mlp.fc1.weight") - 0.02) < 0.001
            assert abs(sd[f"blocks.{i}.{h}_mlp.fc2.weight"].mean().item()) < 1e-3
    assert abs(std_of("t_embedder.mlp.0.weight") - 0.01) < 0.001 and abs(std_of("t_embedder.mlp.2.weight") - 0.01) < 0.001
    assert abs(std_of("final_layer.adaLN_modulation.1.weight") - 0.01) < 0.001
    assert abs(std_of("final_layer.linear.weight") - 0.001) < 0.0002
    assert abs(std_of("x_embedder.proj.weight") - 0.02) < 0.002 and abs(std_of("external_cond.weight") - 0.02) < 0.002
    # with the adaLN gates zeroed every block is the identity: zeroing a block's QKV / fc1 weights cannot change the output
    # (SURVEY.md a13 probe) — checked on the oracle with this very state dict
    cfg = O.DiTConfig(input_h=8, input_w=16, hidden_size=256, depth=2, num_heads=4)
    g = torch.Generator().manual_seed(1)
    x, t = torch.randn(1, 2, 16, 8, 16, generator=g), torch.tensor([[15, 400]])
    sd0 = {k: v for k, v in sd.items() if not k.endswith("rotary_emb.freqs")}
    sd1 = dict(sd0)
    sd1["blocks.0.s_attn.to_qkv.weight"] = torch.zeros_like(sd0["blocks.0.s_attn.to_qkv.weight"])
    sd1["blocks.1.t_mlp.fc1.weight"] = torch.zeros_like(sd0["blocks.1.t_mlp.fc1.weight"])
    with torch.no_grad():
        assert torch.equal(O.dit_forward(sd0, cfg, x, t, None), O.dit_forward(sd1, cfg, x, t, None))
    v = AutoencoderKL(latent_dim=16, input_height=64, input_width=96, patch_size=8, enc_dim=128, enc_depth=1, enc_heads=2, dec_dim=128,
                      dec_depth=1, dec_heads=2)
    vs = v.state_dict()
    w = vs["encoder.0.mlp.fc1.weight"]                        # (512, 128): xavier bound sqrt(6 / (128 + 512))
    bound = (6.0 / (128 + 512)) ** 0.5
    assert w.abs().max().item() <= bound and abs(w.std().item() - bound / 3 ** 0.5) < 0.003
    assert torch.equal(vs["encoder.0.norm1.weight"], torch.ones(128)) and vs["encoder.0.norm1.bias"].abs().max().item() == 0
    assert vs["predictor.bias"].abs().max().item() == 0


# ------------------------------------------------------------------------------------------------------------------------
# G1: per-op vectors captured from the reference's own sub-modules (tools/make_golden.py g1_per_op) — every functional piece of
# the oracle against the module it restates, on the module's own inputs
# ------------------------------------------------------------------------------------------------------------------------
G1_DIT = dict(input_h=4, input_w=8, patch_size=2, in_channels=16, hidden_size=256, depth=2, num_heads=4, external_cond_dim=25)
SMALL_VAE_KW = dict(latent_dim=16, input_height=64, input_width=96, patch_size=8, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=256,
                    dec_depth=2, dec_heads=4)


def test_g1_dit_ops_match_reference_modules():
    import torch.nn.functional as F
    g = gold("g1_ops_dit.safetensors")
    sd = W.synth_state_dict(W.dit_param_shapes(**G1_DIT), seed=3)
    cfg = O.DiTConfig(**G1_DIT)
    tol = 2e-6
    # PatchEmbed (model/dit.py:38-76)
    y = O.patch_embed(g["x_embedder_in0"], sd["x_embedder.proj.weight"], sd["x_embedder.proj.bias"], 2)
    assert rel(y, g["x_embedder_out"]) < tol
    # TimestepEmbedder (model/dit.py:79-123)
    e = O.timestep_embedding(g["t_embedder_in0"], 256)
    te = F.linear(F.silu(F.linear(e, sd["t_embedder.mlp.0.weight"], sd["t_embedder.mlp.0.bias"])), sd["t_embedder.mlp.2.weight"],
                  sd["t_embedder.mlp.2.bias"])
    assert rel(te, g["t_embedder_out"]) < tol
    # SpatialAxialAttention (model/attention.py:73-136), axial RoPE of the 2 x 4 token grid
    s_angles = O.rope_angles_axial(2, 4, O.rope_freqs_pixel(32, 256))
    assert rel(O.spatial_attention(sd, "blocks.0.s_attn.", g["s_attn_in0"], 4, s_angles), g["s_attn_out"]) < tol
    # TemporalAxialAttention (model/attention.py:13-71), causal, windows of 2 .. 5 frames
    t_freqs = O.rope_freqs_lang(64)
    for T in (2, 3, 4, 5):
        assert rel(O.temporal_attention(sd, "blocks.0.t_attn.", g[f"T{T}_t_attn_in"], 4, t_freqs), g[f"T{T}_t_attn_out"]) < tol, T
    # timm Mlp with tanh-GELU
    assert rel(O.mlp(sd, "blocks.0.s_mlp.", g["s_mlp_in0"], "tanh"), g["s_mlp_out"]) < tol
    # a whole SpatioTemporalDiTBlock: (x, c) -> x (model/dit.py:200-225)
    assert rel(O.dit_block(sd, cfg, 1, g["block1_in0"], g["block1_in1"], s_angles, t_freqs), g["block1_out"]) < tol
    # FinalLayer (model/dit.py:126-145): adaLN (shift, scale), LN, modulate, Linear
    x, c = g["final_layer_in0"], g["final_layer_in1"]
    m = F.linear(F.silu(c), sd["final_layer.adaLN_modulation.1.weight"], sd["final_layer.adaLN_modulation.1.bias"])
    shift, scale = m.chunk(2, dim=-1)
    fl = F.linear(O.modulate(O._ln(x), shift, scale), sd["final_layer.linear.weight"], sd["final_layer.linear.bias"])
    assert rel(fl, g["final_layer_out"]) < tol
    # and the composition
    with torch.no_grad():
        assert rel(O.dit_forward(sd, cfg, g["x"], g["t"], g["actions"]), g["out"]) < 1e-5


def test_g1_vae_ops_match_reference_modules():
    g = gold("g1_ops_vae.safetensors")
    sd = W.synth_state_dict(W.vae_param_shapes(**SMALL_VAE_KW), seed=5)
    cfg = O.VAEConfig(**SMALL_VAE_KW)
    ang = O.vae_rope_angles(cfg, cfg.enc_heads, cfg.enc_dim)
    assert rel(O.vae_attention(sd, "encoder.0.attn.", g["attn_in"], cfg.enc_heads, ang, cfg.seq_h, cfg.seq_w), g["attn_out"]) < 2e-6
    assert rel(O.vae_block(sd, "encoder.0.", g["block_in"], cfg.enc_heads, ang, cfg.seq_h, cfg.seq_w), g["block_out"]) < 2e-6
    # patchify / unpatchify (model/vae.py:258-304): the oracle's unpatchify inverts the reference's patchify, the product's host mirror
    # patchifies like the reference
    assert torch.equal(O.vae_unpatchify(g["patchify"], cfg), g["img"])
    assert torch.equal(g["unpatchify_of_patchify"], g["img"])
    from gtav_amd.model.vae import AutoencoderKL
    mirror = AutoencoderKL(**SMALL_VAE_KW, init_weights=False)         # host-side layout helpers only: no GPU call
    assert torch.equal(mirror.patchify(g["img"]), g["patchify"]) and torch.equal(mirror.unpatchify(g["patchify"]), g["img"])
    with torch.no_grad():
        assert rel(O.vae_encode_mean(sd, cfg, g["img"]), g["mean"]) < 1e-5


# ------------------------------------------------------------------------------------------------------------------------
# G8: the reference's own training step (autograd through model/dit.py, clip_grad_norm_, torch.optim.AdamW) — pins the
# training oracle (oracle.dit_loss_and_grads / adamw_reference) to the real module
# ------------------------------------------------------------------------------------------------------------------------
SMALL_DIT_KW = dict(input_h=8, input_w=16, patch_size=2, in_channels=16, hidden_size=256, depth=2, num_heads=4, external_cond_dim=25)


def _g8_sel(v):
    return v.reshape(-1) if v.numel() <= 4096 else v.reshape(-1)[::97]


def test_g8_training_oracle_matches_reference_autograd_and_adamw():
    g = gold("g8_training.safetensors")
    sd = W.synth_state_dict(W.dit_param_shapes(**SMALL_DIT_KW), seed=3)
    loss, v_pred, grads = O.dit_loss_and_grads(sd, O.DiTConfig(**SMALL_DIT_KW), g["x"], g["t"], g["actions"], g["v_target"])
    assert rel(v_pred, g["v_pred"]) < 1e-5 and abs(float(loss) - float(g["loss"])) < 1e-6
    names = sorted(grads)
    assert len(names) == int(g["names_check"])
    norms = torch.stack([grads[k].norm() for k in names])
    assert rel(norms, g["grad_norms"]) < 1e-4
    for k in names:
        assert rel(_g8_sel(grads[k]), g["grad." + k]) < 2e-4, k
    params = {k: v for k, v in sd.items() if not k.endswith("freqs")}
    after, total = O.adamw_reference(params, grads, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0, steps=1)
    assert abs(float(total) - float(g["total_grad_norm"])) < 1e-4 * float(g["total_grad_norm"])
    for k in names:
        upd_ref = g["after." + k] - _g8_sel(sd[k])
        upd = _g8_sel(after[k]) - _g8_sel(sd[k])
        # first Adam step: |update| ~ lr per element, sign(g): elements whose reference gradient is ~0 may flip; compare where it is not
        big = _g8_sel(grads[k]).abs() > 1e-3 * _g8_sel(grads[k]).abs().max()
        assert rel(upd[big], upd_ref[big]) < 1e-3, k


def test_oracle_resize_against_g11():
    """Fixture G11 (tools/make_golden.py g11_resize): `transforms.Resize((360, 640))` of generate.py:150-153 / web_dataset.py:105-107 restated against torch —
    the oracle's resize_frames must reproduce the committed output rows of the three probe images."""
    from safetensors.torch import load_file
    from helpers import G11_ROWS, G11_SIZES, resize_probe_image
    g = load_file(os.path.join(GOLD, "g11_resize.safetensors"))
    for tag, (H, W_) in G11_SIZES.items():
        img = resize_probe_image(H, W_).permute(2, 0, 1)[None].float() / 255.0
        y = O.resize_frames(img)[0]
        assert (y[:, list(G11_ROWS)] - g[f"{tag}.rows"]).abs().max().item() < 1e-6, tag
        assert (y[:, ::8, ::8] - g[f"{tag}.stride8"]).abs().max().item() < 1e-6, tag
