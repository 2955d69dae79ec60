"""Generates tests/golden/*.safetensors by running the ACTUAL reference (/root/reference) on CPU.

Run in the build container only:   PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
The reference is imported through tools/ref_shim.py (stubs for the absent third-party packages); its
modules are loaded with this repo's deterministic synthetic weights (gtav_amd.weights.synth_state_dict, every
matrix non-zero — the reference's own init zeroes the adaLN gates, which would make every block the identity).
Only inputs and reference OUTPUTS are stored (data, not source).  Loops that live inside un-importable scripts
(generate.py:main needs CUDA; DiffusionTrainer needs accelerate state) are driven here with the reference's own
`denoise_step`, `sigmoid_beta_schedule` and model classes, following generate.py:186-220 / train_dit.py:554-650.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.dont_write_bytecode = True

import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

import ref_shim  # noqa: E402
import gtav_amd.weights as W  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SMALL_DIT = dict(input_h=8, input_w=16, patch_size=2, in_channels=16, hidden_size=256, depth=2, num_heads=4, external_cond_dim=25)
SMALL_VAE = dict(latent_dim=16, input_height=64, input_width=96, patch_size=8, enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=256,
                 dec_depth=2, dec_heads=4)


def load_into(model, sd):
    msd = model.state_dict()
    for k in msd:
        if k in sd:
            assert msd[k].shape == sd[k].shape, k
            msd[k] = sd[k]
    model.load_state_dict(msd)
    return model.eval()


def save(name, tensors):
    tensors = {k: (v.detach().contiguous() if torch.is_tensor(v) else torch.tensor(v)) for k, v in tensors.items()}
    path = os.path.join(OUT, name)
    save_file(tensors, path)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, {len(tensors)} tensors")


def one_hot_actions(B, T, g):
    a = torch.zeros(B, T, 25)
    a[torch.arange(B)[:, None], torch.arange(T)[None], torch.randint(0, 25, (B, T), generator=g)] = 1
    return a


SMALL_DIT_NATIVE = dict(input_h=18, input_w=32, patch_size=2, in_channels=16, hidden_size=256, depth=2, num_heads=4, external_cond_dim=25)


@torch.no_grad()
def g7_harness(rd, rv, ref_denoise_step, ref_schedule, full_dit=False):
    """G9 (full_dit=True): the same harness with the PRODUCTION-SIZE DiT-S/2 (depth 16, hidden 1024, 608 M parameters, the repo's seed-0
    synthetic weights) — BASELINE configs[0] exactly: 33 full-size forwards with windows of 2, 3 and 4 frames, full-size VAE on both ends.
    G7: BASELINE config 1 end to end through the reference's own harness functions — dummy-ramp first frame
    (dummy_dataset.py:15-36) -> generate.vae_encode (generate.py:50-66, full-size ViT-L/20 VAE) -> 4-frame / 10-step rollout
    with train_dit.denoise_step (generate.py:186-220; small DiT on the native 18x32 latent grid, action "W") -> decode tail
    (generate.py:238-244: vae.decode(x / s), (x + 1) / 2, clamp(x * 255, 0, 255).byte()).  Stored: inputs, latents after the
    encode and after the rollout, the float frames before the byte conversion and the uint8 frames, both spatially decimated
    (the full tensors are 11 MB; every 4th row / column keeps all 576 patches and every in-patch row phase)."""
    from dummy_dataset import ImageDataset as RefDummy
    import generate as ref_generate                      # module level is import-safe; main() needs CUDA and is not called
    from einops import rearrange
    vsd = W.synth_state_dict(W.vae_param_shapes(), seed=1)
    vae = load_into(rv.VAE_models["vit-l-20-shallow-encoder"](), vsd)
    if full_dit:
        sd = W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0)
        dit = load_into(rd.DiT_models["DiT-S/2"](), sd)
    else:
        sd = W.synth_state_dict(W.dit_param_shapes(**SMALL_DIT_NATIVE), seed=31)
        dit = load_into(rd.DiT(**SMALL_DIT_NATIVE), sd)
    clip = RefDummy(split="test")[0]["video"]            # (5, 3, 360, 640) blue -> red ramp
    n_prompt, total, steps = 1, 4, 10
    prompt = clip[None, :n_prompt].contiguous()          # --start_frame convention: the first frame is the prompt
    x = ref_generate.vae_encode(prompt, vae, n_prompt)   # (1, 1, 16, 18, 32)
    lat_prompt = x.clone()
    g = torch.Generator().manual_seed(77)
    noise = torch.randn(1, total - n_prompt, 16, 18, 32, generator=g)
    actions = torch.zeros(1, total, 25)
    actions[:, :, 3] = 1                                  # generate.py:159,181
    ac = torch.cumprod(1.0 - ref_schedule(1000, clamp_min=0.0001).float(), dim=0)[:, None, None, None]   # generate.py:195-198
    nr = torch.linspace(0, 999, steps + 1)                # generate.py:194
    for i in range(n_prompt, total):
        chunk = torch.clamp(noise[:, i - n_prompt: i - n_prompt + 1], -20, 20)
        x = torch.cat([x, chunk], dim=1)
        start = max(0, i + 1 - dit.max_frames)
        for noise_idx in reversed(range(0, steps + 1)):
            xp, _ = ref_denoise_step(dit, x, actions, noise_idx, 15, nr, ac, start_frame=start, dtype=torch.bfloat16)
            x[:, -1:] = xp[:, -1:]
    lat_final = x.clone()
    z = rearrange(x, "b t c h w -> (b t) (h w) c")
    img = (vae.decode(z / 0.07843137255) + 1) / 2
    img = rearrange(img, "(b t) c h w -> b t h w c", t=total)
    u8 = torch.clamp(img * 255, 0, 255).byte()
    inside = ((img * 255 > 0) & (img * 255 < 255)).float().mean().item()
    tag = "g9" if full_dit else "g7"
    print(f"{tag}: {inside * 100:.1f} % of the float pixels are strictly inside (0, 255); u8 mean {u8.float().mean():.1f}")
    save("g9_config0_full.safetensors" if full_dit else "g7_harness.safetensors", {"prompt_frames": prompt[:, :, :, ::8, ::8].clone(), "noise": noise, "actions": actions,
                                    "latents_prompt": lat_prompt, "latents_final": lat_final,
                                    "frames_f32_stride4": img[:, :, ::4, ::4].clone(), "frames_u8_stride4": u8[:, :, ::4, ::4].clone(),
                                    "frames_u8_row100": u8[:, :, 100].clone()})


@torch.no_grad()
def g1_per_op(rd, rv):
    """G1: per-op input / output pairs of the reference's OWN sub-modules, captured with forward hooks while the
    small reference models run (weights = synth_state_dict seeds 3 / 5 as in G2, so the tests rebuild them): PatchEmbed,
    TimestepEmbedder, SpatialAxialAttention, TemporalAxialAttention (windows of 2..5 frames), timm Mlp, one whole
    SpatioTemporalDiTBlock (x, c -> x), FinalLayer (before unpatchify), and of the VAE: Attention, AttentionBlock, patchify /
    unpatchify."""
    cfg = dict(SMALL_DIT, input_h=4, input_w=8)          # 2 x 4 = 8 tokens per frame: keeps every tensor small
    sd = W.synth_state_dict(W.dit_param_shapes(**cfg), seed=3)
    m = load_into(rd.DiT(**cfg), sd)
    out = {}
    caps = {}

    def hook(name):
        def f(mod, inp, o):
            caps[name] = ([i.detach().clone() for i in inp if torch.is_tensor(i)], o.detach().clone())
        return f
    names = {"x_embedder": m.x_embedder, "t_embedder": m.t_embedder, "s_attn": m.blocks[0].s_attn, "t_attn": m.blocks[0].t_attn,
             "s_mlp": m.blocks[0].s_mlp, "block1": m.blocks[1], "final_layer": m.final_layer}
    hs = [mod.register_forward_hook(hook(n)) for n, mod in names.items()]
    g = torch.Generator().manual_seed(21)
    with torch.no_grad():
        for T in (2, 3, 4, 5):
            x = torch.randn(1, T, 16, 4, 8, generator=g)
            t = torch.randint(0, 1000, (1, T), generator=g)
            a = one_hot_actions(1, T, g)
            y = m(x, t, a)
            out[f"T{T}_t_attn_in"], out[f"T{T}_t_attn_out"] = caps["t_attn"][0][0], caps["t_attn"][1]
            if T == 3:
                out.update({"x": x, "t": t, "actions": a, "out": y})
                for n in names:
                    if n == "t_attn":
                        continue
                    ins, o = caps[n]
                    for j, v in enumerate(ins):
                        out[f"{n}_in{j}"] = v
                    out[f"{n}_out"] = o
    for h in hs:
        h.remove()
    save("g1_ops_dit.safetensors", out)

    vsd = W.synth_state_dict(W.vae_param_shapes(**SMALL_VAE), seed=5)
    v = load_into(rv.AutoencoderKL(**SMALL_VAE), vsd)
    vout, caps2 = {}, {}
    hs = [v.encoder[0].attn.register_forward_hook(lambda mod, inp, o: caps2.__setitem__("attn", (inp[0].detach().clone(), o.detach().clone()))),
          v.encoder[0].register_forward_hook(lambda mod, inp, o: caps2.__setitem__("block", (inp[0].detach().clone(), o.detach().clone())))]
    g = torch.Generator().manual_seed(22)
    img = torch.rand(1, 3, 64, 96, generator=g) * 2 - 1
    with torch.no_grad():
        post = v.encode(img)
        pt = v.patchify(img)
        vout.update({"img": img, "mean": post.mean, "patchify": pt, "unpatchify_of_patchify": v.unpatchify(pt),
                     "attn_in": caps2["attn"][0], "attn_out": caps2["attn"][1], "block_in": caps2["block"][0], "block_out": caps2["block"][1]})
    for h in hs:
        h.remove()
    save("g1_ops_vae.safetensors", vout)


def g8_training(rd):
    """G8: what `accelerator.backward` + `AdamW.step` produce on the REFERENCE module itself — torch autograd through
    model/dit.py's DiT (train_dit.py:649-650, :680), clip_grad_norm_ 1.0 and torch.optim.AdamW(betas (0.9, 0.999), eps 1e-7, wd 0.01)
    (:232-238, :965-968).  Small DiT (seed 3), B = 2, T = 3.  Stored: inputs, v_pred, loss, the gradient norm of EVERY parameter (sorted
    by name), complete gradients of the small tensors, every 97th element of the large ones, and the same selection of the weights after
    one optimizer step."""
    sd = W.synth_state_dict(W.dit_param_shapes(**SMALL_DIT), seed=3)
    m = load_into(rd.DiT(**SMALL_DIT), sd).train()
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 3, 16, 8, 16, generator=g)
    t = torch.randint(0, 1000, (2, 3), generator=g)
    a = one_hot_actions(2, 3, g)
    vt = torch.randn(2, 1, 16, 8, 16, generator=g)
    params = {k: p for k, p in m.named_parameters() if p.requires_grad}
    opt = torch.optim.AdamW(list(params.values()), lr=1e-3, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-7)
    v_pred = m(x, t, a)
    loss = torch.nn.functional.mse_loss(v_pred[:, -1:], vt)
    loss.backward()
    names = sorted(k for k in params if params[k].grad is not None)
    out = {"x": x, "t": t, "actions": a, "v_target": vt, "v_pred": v_pred.detach(), "loss": loss.detach().reshape(1),
           "grad_norms": torch.stack([params[k].grad.norm() for k in names])}
    sel = lambda v: v.reshape(-1).clone() if v.numel() <= 4096 else v.reshape(-1)[::97].clone()
    for k in names:
        out["grad." + k] = sel(params[k].grad)
    total = torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)
    out["total_grad_norm"] = total.reshape(1)
    opt.step()
    for k in names:
        out["after." + k] = sel(params[k].detach())
    out["names_check"] = torch.tensor([len(names)])
    save("g8_training.safetensors", out)
    print("G8 parameters with gradients:", len(names), "; without:", [k for k in params if params[k].grad is None])


@torch.no_grad()
def g10_frame_loop(rd, ref_schedule):
    """G10: the frame loop of `_shared_step` (train_dit.py:590-682) on clips with THREE target frames (7 frames, 4 prompt): per target
    frame i the reference's statements — window slice, timesteps from the per-frame noise indices, noising of context and target frame,
    v-target, `self.dit(x_noisy, t, actions_curr)`, `mse_loss(v_pred[:, -1:], v_target)` — with the draws injected, on the reference DiT
    module (small width).  Stored: inputs, per-frame losses, the returned mean loss, v_pred of every target frame."""
    sd = W.synth_state_dict(W.dit_param_shapes(**SMALL_DIT), seed=17)
    m = load_into(rd.DiT(**SMALL_DIT), sd)
    g = torch.Generator().manual_seed(14)
    B, total, n_prompt, max_frames = 2, 7, 4, 5
    lat = torch.randn(B, total, 16, 8, 16, generator=g) * 0.5
    actions = one_hot_actions(B, total, g)
    n_iter = total - n_prompt
    tgt = torch.randint(1, 51, (n_iter, B), generator=g)
    ctx = torch.randint(1, 41, (n_iter, B), generator=g)
    ctx_m = torch.minimum(ctx, tgt)
    ac = torch.cumprod(1.0 - ref_schedule(1000, clamp_min=0.000001).float(), dim=0)[:, None, None, None]
    nr = torch.linspace(0, 999, 51).long()
    out = {"latents": lat, "actions": actions, "target_idx": tgt, "ctx_idx": ctx}
    total_loss = 0.0
    for idx, i in enumerate(range(n_prompt, total)):
        x_input, a_input = lat[:, : i + 1], actions[:, : i + 1]
        start = max(0, i + 1 - max_frames)
        t = torch.zeros((B, i + 1), dtype=torch.long)
        t[:, :-1] = nr[ctx_m[idx]].unsqueeze(1)
        t[:, -1] = nr[tgt[idx]]
        x_curr, t, a_curr = x_input[:, start:], t[:, start:], a_input[:, start:]
        cn = torch.randn(x_curr[:, :-1].shape, generator=g) * 6
        nz = torch.randn(x_curr[:, -1:].shape, generator=g) * 6
        out[f"ctx_noise{idx}"], out[f"noise{idx}"] = cn.clone(), nz.clone()
        cn, nz = cn.clamp(-20, 20), nz.clamp(-20, 20)
        x_noisy = x_curr.clone()
        al = ac[t[:, :-1]]
        x_noisy[:, :-1].mul_(al.sqrt()).add_((1 - al).sqrt() * cn)
        al = ac[t[:, -1:]]
        x_noisy[:, -1:].mul_(al.sqrt()).add_((1 - al).sqrt() * nz)
        v_target = al.sqrt() * nz - (1 - al).sqrt() * x_curr[:, -1:]
        v_pred = m(x_noisy, t, a_curr)
        loss = torch.nn.functional.mse_loss(v_pred[:, -1:], v_target)
        total_loss = total_loss + loss
        out[f"loss{idx}"], out[f"v_pred{idx}"], out[f"v_target{idx}"] = loss.reshape(1), v_pred, v_target
    out["mean_loss"] = (total_loss / n_iter).reshape(1)
    save("g10_frame_loop.safetensors", out)


def resize_probe_image(H, W):
    """A deterministic uint8 image (H, W, 3) from integer arithmetic only (no RNG: the tests rebuild it bit for bit): smooth ramps + a fine texture."""
    y = torch.arange(H).view(H, 1, 1)
    x = torch.arange(W).view(1, W, 1)
    c = torch.arange(3).view(1, 1, 3)
    return ((x * 7 + y * 13 + c * 29 + (x * y) % 11 + ((x // 3) * (y // 5)) % 17 * 9) % 256).to(torch.uint8)


G11_ROWS = (0, 1, 100, 179, 180, 358, 359)


@torch.no_grad()
def g11_resize():
    """G11: frame I/O on either side of the path — `transforms.Resize((360, 640))` as generate.py:150-153 applies it to the --start_frame
    image and web_dataset.py:105-107 / hf_dataset.py:30-33 to every dataset frame.  torchvision is absent from the image, so its semantics are restated
    against torch: transforms.Resize on a float tensor is torchvision.transforms.functional.resize -> torch.nn.functional.interpolate(mode="bilinear",
    align_corners=False, antialias=True) (torchvision >= 0.17 default; README.md:65 installs the newest).  Two probes: a 500 x 333 (W x H) image
    (mixed up- / down-scaling: the antialias filter is live along x... and y) and the dataset's own 480 x 270 frame size (pure up-scaling).
    Stored: seven full-width output rows and a stride-8 subsample of each result (the input is rebuilt in the tests from resize_probe_image)."""
    import torch.nn.functional as F
    out = {}
    for tag, (H, W_) in (("500x333", (333, 500)), ("480x270", (270, 480)), ("1280x720", (720, 1280))):
        img = resize_probe_image(H, W_).permute(2, 0, 1)[None].float() / 255.0          # ToTensor()
        y = F.interpolate(img, size=(360, 640), mode="bilinear", align_corners=False, antialias=True)
        out[f"{tag}.rows"] = y[0][:, list(G11_ROWS)].clone()
        out[f"{tag}.stride8"] = y[0][:, ::8, ::8].clone()
    save("g11_resize.safetensors", out)


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--only-g11" in sys.argv:
        return g11_resize()
    rd, rv, ref_denoise_step, ref_schedule = ref_shim.import_reference()
    g11_resize()
    if "--only-g9-g10" in sys.argv:
        g10_frame_loop(rd, ref_schedule)
        return g7_harness(rd, rv, ref_denoise_step, ref_schedule, full_dit=True)
    if "--only-g7" in sys.argv:
        return g7_harness(rd, rv, ref_denoise_step, ref_schedule)
    if "--only-g1-g8" in sys.argv:
        g1_per_op(rd, rv)
        return g8_training(rd)
    from model.rotary_embedding_torch import RotaryEmbedding
    from dummy_dataset import ImageDataset as RefDummy
    from web_dataset import actions_to_one_hot as ref_one_hot

    # ---------------- G0: constants / tables -------------------------------------------------
    g0 = {}
    for tag, cm in (("gen", 1e-4), ("train", 1e-6)):
        betas = ref_schedule(1000, clamp_min=cm)
        g0[f"betas64_{tag}"] = betas
        g0[f"alphas_cumprod_{tag}"] = torch.cumprod(1.0 - betas.float(), dim=0)
    g0["noise_range_gen100"] = torch.linspace(0, 999, 101)
    g0["noise_range_gen100_long"] = torch.tensor([int(torch.full((1,), v, dtype=torch.long)) for v in torch.linspace(0, 999, 101)])
    g0["noise_range_train50"] = torch.linspace(0, 999, 51).long()
    sp = RotaryEmbedding(dim=32, freqs_for="pixel", max_freq=256)
    g0["rope_spatial_freqs"] = sp.freqs.detach()
    g0["rope_spatial_angles_9x16"] = sp.get_axial_freqs(9, 16)
    tp = RotaryEmbedding(dim=64)
    g0["rope_temporal_freqs"] = tp.freqs.detach()
    g0["rope_temporal_angles_T5"] = tp.forward(torch.arange(5).float(), tp.freqs, seq_len=5).clone()
    g0["rope_vae_angles_18x32"] = RotaryEmbedding(dim=16, freqs_for="pixel", max_freq=18 * 32).get_axial_freqs(18, 32)
    g0["timestep_embedding_rows"] = rd.TimestepEmbedder.timestep_embedding(torch.tensor([0, 15, 19, 500, 999]), 256)
    g0["modulate_known"] = rd.modulate(torch.ones(1, 1, 1, 1, 1), torch.full((1, 1, 1), 0.5), torch.full((1, 1, 1), 0.25))
    ds = RefDummy(split="test")
    g0["dummy_clip_means"] = ds.sequence_blue_red.mean(dim=(2, 3))            # (5, 3) colours
    g0["one_hot_example"] = ref_one_hot([-1, 3, 0, 24, -1])
    save("g0_constants.safetensors", g0)

    # ---------------- G2: small DiT / VAE end to end -----------------------------------------
    sd = W.synth_state_dict(W.dit_param_shapes(**SMALL_DIT), seed=3)
    m = load_into(rd.DiT(**SMALL_DIT), sd)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 3, 16, 8, 16, generator=g)
    t = torch.randint(0, 1000, (2, 3), generator=g)
    a = one_hot_actions(2, 3, g)
    taps = {}
    hooks = [m.blocks[i].register_forward_hook(lambda mod, inp, out, i=i: taps.__setitem__(i, out.clone())) for i in range(2)]
    out_a = m(x, t, a)
    blk = {f"block{i}_actions": taps[i] for i in range(2)}
    out_n = m(x, t, None)
    for h in hooks:
        h.remove()
    save("g2_small_dit.safetensors", {"x": x, "t": t, "actions": a, "out_actions": out_a, "out_noactions": out_n, **blk})

    vsd = W.synth_state_dict(W.vae_param_shapes(**SMALL_VAE), seed=5)
    v = load_into(rv.AutoencoderKL(**SMALL_VAE), vsd)
    g = torch.Generator().manual_seed(1)
    img = torch.rand(3, 3, 64, 96, generator=g) * 2 - 1
    post = v.encode(img)
    z = torch.randn(3, v.seq_len, 16, generator=g)
    save("g2_small_vae.safetensors", {"img": img, "mean": post.mean, "logvar": post.logvar, "z": z, "decoded": v.decode(z)})

    # ---------------- G4: denoise_step (reference function, small DiT) -----------------------
    sd4 = W.synth_state_dict(W.dit_param_shapes(**SMALL_DIT), seed=4)
    m4 = load_into(rd.DiT(**SMALL_DIT), sd4)
    g = torch.Generator().manual_seed(2)
    x4 = torch.randn(2, 6, 16, 8, 16, generator=g)
    a4 = torch.zeros(2, 6, 25)
    a4[:, :, 3] = 1
    ac = torch.cumprod(1.0 - ref_schedule(1000).float(), dim=0)[:, None, None, None]
    nr = torch.linspace(0, 999, 11)
    g4 = {"x": x4, "actions": a4}
    for idx in (10, 4, 0):
        xp, vp = ref_denoise_step(m4, x4, a4, idx, 15, nr, ac, start_frame=1, dtype=torch.bfloat16)
        g4[f"x_pred_{idx}"], g4[f"v_pred_{idx}"] = xp, vp
    save("g4_denoise_step.safetensors", g4)

    # ---------------- G5: config-1-shaped mini rollout (generate.py:186-220) ------------------
    sd5 = W.synth_state_dict(W.dit_param_shapes(**SMALL_DIT), seed=6)
    m5 = load_into(rd.DiT(**SMALL_DIT), sd5)
    g = torch.Generator().manual_seed(9)
    B, total, steps = 2, 4, 10
    x0 = torch.randn(B, 1, 16, 8, 16, generator=g) * 0.5
    noise = torch.randn(B, 3, 16, 8, 16, generator=g)
    a5 = torch.zeros(B, total, 25)
    a5[:, :, 3] = 1
    xx = x0.clone()
    nr5 = torch.linspace(0, 999, steps + 1)
    for i in range(1, total):
        chunk = torch.clamp(noise[:, i - 1: i], -20, 20)
        xx = torch.cat([xx, chunk], dim=1)
        start = max(0, i + 1 - 5)
        for noise_idx in reversed(range(0, steps + 1)):
            xp, _ = ref_denoise_step(m5, xx, a5, noise_idx, 15, nr5, ac, start_frame=start, dtype=torch.bfloat16)
            xx[:, -1:] = xp[:, -1:]
    save("g5_rollout.safetensors", {"x_prompt": x0, "noise": noise, "actions": a5, "latents": xx})

    # ---------------- G6: training forward + loss (train_dit.py:574-650) ----------------------
    sd6 = W.synth_state_dict(W.dit_param_shapes(**SMALL_DIT), seed=7)
    m6 = load_into(rd.DiT(**SMALL_DIT), sd6)
    g = torch.Generator().manual_seed(4)
    B = 3
    lat = torch.randn(B, 5, 16, 8, 16, generator=g) * 0.5
    a6 = torch.zeros(B, 5, 25)
    a6[:, -1, 1] = 1
    tgt, ctx = torch.tensor([50, 1, 23]), torch.tensor([40, 7, 30])
    ctx_noise = torch.randn(B, 4, 16, 8, 16, generator=g) * 8
    nz = torch.randn(B, 1, 16, 8, 16, generator=g) * 8
    ac_t = torch.cumprod(1.0 - ref_schedule(1000, clamp_min=0.000001).float(), dim=0)[:, None, None, None]
    noise_range = torch.linspace(0, 999, 51).long()
    ctx_i = torch.minimum(ctx, tgt)
    tt = torch.zeros((B, 5), dtype=torch.long)
    tt[:, :-1] = noise_range[ctx_i].unsqueeze(1)
    tt[:, -1] = noise_range[tgt]
    cn, n1 = ctx_noise.clamp(-20, 20), nz.clamp(-20, 20)
    x_noisy = lat.clone()
    al = ac_t[tt[:, :-1]]
    x_noisy[:, :-1].mul_(al.sqrt()).add_((1 - al).sqrt() * cn)
    al = ac_t[tt[:, -1:]]
    x_noisy[:, -1:].mul_(al.sqrt()).add_((1 - al).sqrt() * n1)
    v_target = al.sqrt() * n1 - (1 - al).sqrt() * lat[:, -1:]
    v_pred = m6(x_noisy, tt, a6)
    loss = torch.nn.functional.mse_loss(v_pred[:, -1:], v_target)
    save("g6_train_forward.safetensors", {"latents": lat, "actions": a6, "target_idx": tgt, "ctx_idx": ctx, "ctx_noise": ctx_noise,
                                          "noise": nz, "x_noisy": x_noisy, "t": tt, "v_target": v_target, "v_pred": v_pred,
                                          "loss": loss.reshape(1)})

    # ---------------- G3: full-size models (weights regenerated from the hash generator in tests) ----
    fsd = W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0)
    fm = load_into(rd.DiT_models["DiT-S/2"](), fsd)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 5, 16, 18, 32, generator=g)
    t = torch.tensor([[15, 15, 15, 15, 500]])
    a = torch.zeros(1, 5, 25)
    a[:, :, 3] = 1
    taps = {}
    hooks = [fm.blocks[i].register_forward_hook(lambda mod, inp, out, i=i: taps.__setitem__(i, out.clone())) for i in (0, 7, 15)]
    out1 = fm(x, t, a)
    for h in hooks:
        h.remove()
    g3 = {"x_b1t5": x, "t_b1t5": t, "a_b1t5": a, "out_b1t5": out1}
    for i in (0, 7, 15):
        g3[f"block{i}_stats"] = torch.stack([taps[i].mean(), taps[i].abs().max(), taps[i].abs().mean()])
        g3[f"block{i}_samples"] = taps[i].reshape(-1)[:: max(1, taps[i].numel() // 8)][:8].clone()
    g2_ = torch.Generator().manual_seed(5)
    x2 = torch.randn(2, 3, 16, 18, 32, generator=g2_)
    t2 = torch.randint(0, 1000, (2, 3), generator=g2_)
    g3.update({"x_b2t3": x2, "t_b2t3": t2, "out_b2t3": fm(x2, t2, None)})
    save("g3_full_dit.safetensors", g3)
    del fm, fsd

    fvsd = W.synth_state_dict(W.vae_param_shapes(), seed=1)
    fv = load_into(rv.VAE_models["vit-l-20-shallow-encoder"](), fvsd)
    g = torch.Generator().manual_seed(3)
    img = torch.rand(2, 3, 360, 640, generator=g) * 2 - 1       # regenerated from the seed in the tests
    post = fv.encode(img)
    z = torch.randn(2, 576, 16, generator=g)
    dec = fv.decode(z)
    del fv, fvsd
    g7_harness(rd, rv, ref_denoise_step, ref_schedule)
    g10_frame_loop(rd, ref_schedule)
    g7_harness(rd, rv, ref_denoise_step, ref_schedule, full_dit=True)
    g1_per_op(rd, rv)
    g8_training(rd)
    save("g3_full_vae.safetensors", {"mean": post.mean, "logvar": post.logvar, "z": z, "decoded_stride4": dec[:, :, ::4, ::4].clone(),
                                     "decoded_row100": dec[:, :, 100].clone()})


if __name__ == "__main__":
    main()
