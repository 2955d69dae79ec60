#!/usr/bin/env python3
"""Headline benchmark: generated frames/s for a 32-frame clip with 100 noise steps (BASELINE.json metric).

One "step" = one complete clip generation per GPU: VAE-encode 4 prompt frames, 28 generated frames x 101
denoise steps through the DiT, all-gather of the latents across ranks (RCCL over xGMI when N > 1) and
VAE-decode of the 32 frames.  N = 1 runs BASELINE configs[1] (DiT without actions, batch 1); for N > 1 each
rank generates its own `--batch-per-gpu` samples (weak scaling, no data-path collective besides the final
all-gather).  Inputs are synthetic, generated on the CPU from fixed seeds and resident in HBM before the
timed region; weights are the repo's deterministic synthetic weights (no checkpoints exist offline).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     — the dominant kernel (fc1 GEMM with the GELU epilogue, csrc/gemm.hip), timed in situ with HIP events on the
                 launch stream during real forwards (gtav_dit_profile; the events are attached to the kernel's own dispatch,
                 hipExtLaunchKernel), algorithmic FLOPs per launch / mean duration
  cpu_baseline — the CPU oracle (oracle/ref_cpu.py, fp32 torch CPU kernels — the reference's own CPU path) on the
                 host cores, a bounded sample extrapolated to the clip (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

P_TOK, D_MODEL, DEPTH, HM = 144, 1024, 16, 4096
# SURVEY.md §8(d) geometry presets.  native: the only geometry the reference's factories support (360x640 frames, VAE patch 20 ->
# 16x18x32 latents, 144 DiT tokens per frame).  g256: the literal reading of BASELINE.json's "256x256" (VAE patch 16 -> 16x16x16
# latents, 64 DiT tokens per frame, 256 VAE tokens) built through the reference's constructors with ViT-L/DiT-S widths.
GEOM = {
    "native": dict(frame=(360, 640), lat=(18, 32), vae_tokens=576, vae_gflop=(96.6, 191.6),
                   dit=dict(), vae=dict()),
    "g256": dict(frame=(256, 256), lat=(16, 16), vae_tokens=256, vae_gflop=(40.7, 80.9),
                 dit=dict(input_h=16, input_w=16, patch_size=2, in_channels=16, hidden_size=1024, depth=16, num_heads=16,
                          external_cond_dim=25),
                 vae=dict(latent_dim=16, input_height=256, input_width=256, patch_size=16, enc_dim=1024, enc_depth=6, enc_heads=16,
                          dec_dim=1024, dec_depth=12, dec_heads=16)),
}
MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0


def dit_forward_flops(tokens, frames_q, frames_k_sum, B):
    """Executed FLOPs of one DiT forward over `tokens` query tokens (SURVEY.md §8(d) formula, 2 FLOPs per MAC):
    dense GEMMs 48 D^2 per token per block, spatial attention 4 P^2 D per frame per block, temporal attention
    4 P D per (query frame, key frame) pair per block; embed/final/conditioning are < 0.5 % and not counted."""
    dense = DEPTH * 48 * D_MODEL * D_MODEL * tokens
    spatial = DEPTH * 4 * P_TOK * P_TOK * D_MODEL * frames_q
    temporal = DEPTH * 4 * P_TOK * D_MODEL * frames_k_sum * B
    return dense + spatial + temporal


def clip_flops(B, n_prompt, total, steps, max_frames, cached):
    fl = 0.0
    for i in range(n_prompt, total):
        T = min(i + 1, max_frames)
        full = dit_forward_flops(B * T * P_TOK, B * T, T * (T + 1) // 2, B)
        one = dit_forward_flops(B * P_TOK, B, T, B)
        fl += (full + steps * one) if cached else (steps + 1) * full
    return fl


def host_cores():
    """Threads for the CPU baseline: the cgroup CPU quota if there is one, else the affinity mask, capped at 32
    (torch's intra-op pool oversubscribes badly when handed every logical CPU of a shared host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 32))


def bench_train(args, world, rank, dev, dist):
    """configs[4]: forward + loss of train_dit.py `_shared_step` (configs/train_dit_actions.yaml: batch 16 per GPU, 4 prompt
    frames + 1 target, ddim_noise_steps 50, ctx_max_noise_idx 40, clamp_min 1e-6) on synthetic 360x640 clips.  A step = one
    batch per GPU: VAE-encode 80 frames -> noise -> one DiT forward (B, T=5) -> v-target MSE; the scalar loss is all-reduced."""
    import gtav_amd.weights as W
    from gtav_amd.model.dit import DiT_models
    from gtav_amd.model.vae import VAE_models
    from gtav_amd.train import encode_frames, forward_loss
    B = args.batch_per_gpu if args.batch_per_gpu > 1 else 16
    dit = DiT_models["DiT-S/2"](init_weights=False, max_batch=B)
    dit.load_state_dict(W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0))
    vae = VAE_models["vit-l-20-shallow-encoder"](init_weights=False, max_frames_per_call=40)
    vae.load_state_dict(W.synth_state_dict(W.vae_param_shapes(), seed=1))
    g = torch.Generator().manual_seed(100 + rank)
    frames = torch.rand(B, 5, 3, 360, 640, generator=g).to(dev)
    actions = torch.zeros(B, 5, 25, device=dev)
    actions[:, :, 3] = 1
    tgt = torch.randint(1, 51, (B,), generator=g)
    ctx = torch.randint(1, 41, (B,), generator=g)
    ctx_noise = torch.randn(B, 4, 16, 18, 32, generator=g).to(dev)
    noise = torch.randn(B, 1, 16, 18, 32, generator=g).to(dev)

    def step():
        lat = encode_frames(vae, frames)
        loss, _, _ = forward_loss(dit, lat, actions, tgt, ctx, ctx_noise, noise)
        if world > 1:
            dist.all_reduce(loss, op=dist.ReduceOp.SUM)
            loss /= world
        return loss

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    if rank == 0:
        flops = B * (dit_forward_flops(5 * P_TOK, 5, 15, 1) + 5 * 96.6e9)   # DiT forward + VAE encode
        print(json.dumps({
            "metric": "training forward+loss samples/sec (5-frame clips, configs/train_dit_actions.yaml shapes)",
            "value": round(world * B * args.steps / el, 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(el / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp16 (fp32 accumulate/residual)", "data": "synthetic", "loss": float(loss.item()),
            "config": {"workload": "BASELINE configs[4]: train_dit.py forward+loss, batch %d per GPU, DiT-S/2 + VAE encode of %d frames" % (B, 5 * B),
                       "global_batch": world * B, "parallelism": "data-parallel x%d (forward only; loss all-reduce)" % world},
            "achieved_tflops_per_gpu": round(flops * args.steps / el / 1e12, 1)}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-per-gpu", type=int, default=1)
    ap.add_argument("--total-frames", type=int, default=32)
    ap.add_argument("--noise-steps", type=int, default=100)
    ap.add_argument("--n-prompt", type=int, default=4)
    ap.add_argument("--use-actions", action="store_true", help="BASELINE configs[2]: action-conditioned DiT")
    ap.add_argument("--algo", choices=["window", "cached", "both"], default="both",
                    help="window = recompute the whole window every noise step (reference behaviour, headline value); "
                         "cached = exact context-K/V-cached variant; both = time both (value = window)")
    ap.add_argument("--geometry", choices=["native", "g256"], default="native",
                    help="native = 360x640 frames (the reference's factories; headline); g256 = 256x256 frames, SURVEY.md 8(d) preset")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vae", action="store_true", help="skip VAE encode/decode (DiT loop only; not the headline)")
    ap.add_argument("--mode", choices=["generate", "train"], default="generate",
                    help="train = BASELINE configs[4]: training forward + loss (train_dit.py:554-650: VAE-encode 5-frame clips, noise, "
                         "one DiT forward over the window, MSE vs the v-target), data-parallel, metric samples/s (not the headline)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback for the product path"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)

    import gtav_amd.weights as W
    from gtav_amd.generate import all_gather_latents, generate_latents, vae_decode_frames, vae_encode
    from gtav_amd.model.dit import DiT_models
    from gtav_amd.model.vae import VAE_models

    if args.mode == "train":
        return bench_train(args, world, rank, dev, dist)
    B = args.batch_per_gpu
    total, n_prompt, steps = args.total_frames, args.n_prompt, args.noise_steps
    # ---- models with deterministic synthetic weights (every matrix non-zero, incl. adaLN) ----
    global P_TOK
    geo = GEOM[args.geometry]
    (FH, FW), (LH, LW) = geo["frame"], geo["lat"]
    P_TOK = (LH // 2) * (LW // 2)
    if args.geometry == "native":
        dit = DiT_models["DiT-S/2"](init_weights=False, max_batch=B)
        dit.load_state_dict(W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0))
    else:
        from gtav_amd.model.dit import DiT
        dit = DiT(**geo["dit"], init_weights=False, max_batch=B)
        dit.load_state_dict(W.synth_state_dict(W.dit_param_shapes(**geo["dit"]), seed=0))
    vae = None
    if not args.no_vae:
        nfc = min(32, max(4, B * 4))
        if args.geometry == "native":
            vae = VAE_models["vit-l-20-shallow-encoder"](init_weights=False, max_frames_per_call=nfc)
            vae.load_state_dict(W.synth_state_dict(W.vae_param_shapes(), seed=1))
        else:
            from gtav_amd.model.vae import AutoencoderKL
            vae = AutoencoderKL(**geo["vae"], init_weights=False, max_frames_per_call=nfc)
            vae.load_state_dict(W.synth_state_dict(W.vae_param_shapes(**geo["vae"]), seed=1))

    # ---- synthetic inputs, indexed by GLOBAL sample id so results do not depend on the sharding ----
    def sample_inputs(gid):
        g = torch.Generator().manual_seed(1000 + gid)
        frames = torch.rand(n_prompt, 3, FH, FW, generator=g)
        noise = torch.randn(total - n_prompt, 16, LH, LW, generator=g)
        return frames, noise

    gids = [rank * B + b for b in range(B)]
    ins = [sample_inputs(g) for g in gids]
    frames = torch.stack([f for f, _ in ins]).to(dev)
    noise = torch.stack([n for _, n in ins]).to(dev)
    actions = None
    if args.use_actions:
        actions = torch.zeros(B, total, 25, device=dev)
        actions[:, :, 3] = 1  # "W" for every frame (generate.py:159,181)
    lat_fallback = torch.randn(B, n_prompt, 16, LH, LW, generator=torch.Generator().manual_seed(7)).to(dev) * 0.5

    def one_clip(cached):
        x0 = vae_encode(frames, vae, n_prompt) if vae is not None else lat_fallback
        x = generate_latents(dit, x0, total, steps, noise, actions, ctx_cache=cached)
        xg = all_gather_latents(x)
        out = vae_decode_frames(x, vae) if vae is not None else x
        return xg, out

    def timed(cached, nsteps, nwarm):
        for _ in range(nwarm):
            one_clip(cached)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            one_clip(cached)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        return el

    results = {}
    algos = ["window", "cached"] if args.algo == "both" else [args.algo]
    for algo in algos:
        el = timed(algo == "cached", args.steps, args.warmup)
        results[algo] = el
    head = "window" if "window" in results else algos[0]
    el = results[head]
    gen_frames = world * B * (total - n_prompt)
    value = gen_frames * args.steps / el

    # ---- roofline: in-situ HIP-event timing of every kernel class over real forwards ----
    roofline, classes = None, None
    g = torch.Generator().manual_seed(3)
    xw = torch.randn(B, 5, 16, LH, LW, generator=g).to(dev)
    tw = torch.tensor([[15, 15, 15, 15, 500]] * B)
    aw = actions[:, :5].contiguous() if actions is not None else None
    for _ in range(2):
        dit(xw, tw, aw)
    torch.cuda.synchronize()
    dit.profile(True)
    nprof = 6
    t0 = time.perf_counter()
    for _ in range(nprof):
        dit(xw, tw, aw)
    torch.cuda.synchronize()
    prof = dit.profile_read()
    dit.profile(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nfw = 10
    for _ in range(nfw):
        dit(xw, tw, aw)
    torch.cuda.synchronize()
    fwd_ms = (time.perf_counter() - t0) / nfw * 1e3
    M = B * 5 * P_TOK
    flops_fc1 = 2.0 * M * HM * D_MODEL
    ev_ms, ev_n = prof.pop("empty_event_pair")
    ev_over_ms = ev_ms / max(ev_n, 1)                  # cost of one HIP-event pair around nothing
    ms_fc1, n_fc1 = prof["gemm_fc1"]
    # single-kernel classes (GEMMs, LayerNorm, attention) are timed with start/stop events attached to the kernel's own dispatch
    # packet (hipExtLaunchKernel), i.e. the kernel's begin-to-end time as rocprofv3 reports it; only `other` (multi-kernel) uses
    # event pairs around the launches, which carry ~2 us of marker overhead each (the empty-pair time is reported for reference)
    avg_fc1_ms = ms_fc1 / max(n_fc1, 1)
    ach = flops_fc1 / (avg_fc1_ms * 1e-3) / 1e12
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")   # HBM bytes per fc1 launch from rocprofv3 --pmc passes (profiles/README.md)
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        traffic = tj.get("fc1_M%d" % M, {}).get("hbm_bytes_per_launch")
    roofline = {"kernel": "fc1 GEMM + GELU-tanh epilogue (csrc/gemm.hip, M=%d N=4096 K=1024, fp16 MFMA)" % M, "bound": "mfma",
                "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                "traffic": traffic, "avg_launch_us": round(avg_fc1_ms * 1e3, 2), "launches_timed": int(n_fc1),
                "flops_per_launch": flops_fc1, "timing": "HIP events attached to the dispatch (hipExtLaunchKernel)",
                "empty_event_pair_us": round(ev_over_ms * 1e3, 2)}
    classes = {k: {"ms_per_forward": round(v[0] / nprof, 4), "launches_per_forward": v[1] // nprof} for k, v in prof.items()}
    fwd_flops = dit_forward_flops(M, B * 5, 15, B)
    step_tflops = fwd_flops / (fwd_ms * 1e-3) / 1e12

    # ---- CPU baseline: oracle on the host cores, bounded sample (rank 0, N = 1) ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_cpu as O
        cores = host_cores()
        torch.set_num_threads(cores)
        sd = dit.state_dict()
        cfg = O.dit_s_2() if args.geometry == "native" else O.DiTConfig(**geo["dit"])
        xc = xw[:1].cpu()
        tc = tw[:1]
        with torch.no_grad():
            t0 = time.perf_counter()
            O.dit_forward(sd, cfg, xc, tc, None)          # warm-up, also bounds the sample on a slow host
            warm = time.perf_counter() - t0
            t0 = time.perf_counter()
            nf = 0
            while nf < 1 or (time.perf_counter() - t0 < 10 and nf < 40 and warm < 10):
                O.dit_forward(sd, cfg, xc, tc, None)
                nf += 1
            t_fwd = (time.perf_counter() - t0) / nf
            t_enc = t_dec = 0.0
            if vae is not None:
                vsd, vcfg = vae.state_dict(), (O.vit_l_20_shallow_encoder() if args.geometry == "native" else O.VAEConfig(**geo["vae"]))
                img = frames[0, :2].cpu() * 2 - 1
                t0 = time.perf_counter()
                O.vae_encode_moments(vsd, vcfg, img)
                t_enc = (time.perf_counter() - t0) / 2
                z = torch.randn(2, geo["vae_tokens"], 16)
                t0 = time.perf_counter()
                O.vae_decode(vsd, vcfg, z)
                t_dec = (time.perf_counter() - t0) / 2
        clip_s = (total - n_prompt) * (steps + 1) * t_fwd + n_prompt * t_enc + total * t_dec
        cpu = {"value": round((total - n_prompt) / clip_s, 5), "unit": "generated frames/s", "cores": cores, "kind": "port",
               "sample": "oracle/ref_cpu.py (fp32 torch CPU): %d DiT forwards B=1 T=5 (%.3f s each) + VAE encode/decode of 2 frames "
                         "(%.3f / %.3f s per frame), extrapolated to %d forwards + %d enc + %d dec" %
                         (nf, t_fwd, t_enc, t_dec, (total - n_prompt) * (steps + 1), n_prompt, total)}

    if rank == 0:
        cfg_name = ("DiT-S/2 (608M) + ViT-L/20 VAE, native 360x640 frames -> 16x18x32 latents" if args.geometry == "native" else
                    "DiT-S/2 widths on a 16x16 latent grid + ViT-L/16 VAE, g256 preset: 256x256 frames -> 16x16x16 latents (SURVEY.md 8(d))")
        line = {
            "metric": "generated frames/sec (32-frame clip, 100 noise steps)", "value": round(value, 4), "unit": "generated frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp16 (fp32 accumulate/residual)",
            "data": "synthetic (seeded CPU-generated prompt frames + noise, hash-seeded random weights incl. adaLN)",
            "config": {"workload": "BASELINE configs[%d]: %s, %d frames (%d prompt), %d noise steps, batch %d per GPU%s" %
                                   (2 if args.use_actions else 1, cfg_name, total, n_prompt, steps, B,
                                    ", action-conditioned" if args.use_actions else ", no actions"),
                       "global_batch": world * B, "algorithm": "window-recompute (reference behaviour)" if head == "window" else "ctx-cached",
                       "parallelism": "batch-sharded x%d, all-gather of latents" % world, "vae_in_timed_region": vae is not None,
                       "geometry": args.geometry},
            "all_frames_per_s": round(world * B * total * args.steps / el, 4),          # B * 32 / wall
            "dit_forwards_per_s": round(world * (total - n_prompt) * (steps + 1) * args.steps / el, 2),   # batched forwards of B samples
            "roofline": roofline, "cpu_baseline": cpu,
            "dit_step": {"forward_ms_B%d_T5" % B: round(fwd_ms, 3), "executed_tflop_per_forward": round(fwd_flops / 1e12, 4),
                         "achieved_tflops": round(step_tflops, 1), "frac_of_mfma_peak": round(step_tflops / MFMA_PEAK_TFLOPS, 4),
                         "kernel_classes": classes},
        }
        for algo, e in results.items():
            fl = clip_flops(B, n_prompt, total, steps, 5, algo == "cached") * world
            line["algo_" + algo] = {"generated_frames_per_s": round(gen_frames * args.steps / e, 4), "ms_per_clip": round(e / args.steps * 1e3, 1),
                                    "executed_pflop_per_clip": round(fl / 1e15, 4), "achieved_tflops_per_gpu": round(fl / world * args.steps / e / 1e12, 1)}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
