#!/usr/bin/env python3
"""Headline benchmark: generated frames/s for a 32-frame clip with 100 noise steps (BASELINE.json metric).

One "step" = one complete clip generation per GPU (gtav_amd.generate.generate_clip): VAE-encode 4 prompt frames, 28 generated
frames x 101 denoise steps through the DiT, all-gather of the latents across ranks (RCCL over xGMI when N > 1) and VAE-decode of
the 32 frames.  The headline line is BASELINE configs[1] (DiT without actions, batch 1 per GPU) at every N — weak scaling, the
per-GPU work is the same at N = 1, 2, 4, 8, no data-path collective besides the final all-gather.  A second, bounded leg runs the
batched configuration the north-star's roofline target refers to: batch 8 per GPU with action conditioning (`config2` at N = 1 =
BASELINE configs[2]; `config3` at N > 1 = 8 N sequences sharded 8 per GPU, which is BASELINE configs[3] at N = 8).
Inputs are synthetic, a function of the GLOBAL sample id (gtav_amd.generate.shard_inputs), generated on the CPU from fixed seeds
and resident in HBM before the timed region; weights are the repo's deterministic synthetic weights (no checkpoints offline).

Launching: `python bench.py --gpus N` with WORLD_SIZE unset starts N rank processes itself (one per GPU, env rendezvous on
127.0.0.1) BEFORE anything touches the GPU in this process, waits for them and exits with their status; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of the ranks.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     — the dominant kernel = the GEMM class (to_qkv / out-proj / fc1 / fc2, csrc/gemm.hip) with the LARGEST time per forward, timed
                 in situ with HIP events attached to the kernel's own dispatch (hipExtLaunchKernel) during real forwards; algorithmic FLOPs
                 per launch / mean duration; plus the FLOP-weighted aggregate over the four GEMM classes, the worst class and every class's
                 fraction; `traffic` / `mfma_busy_pmc` from the committed rocprofv3 --pmc passes on the same kernel source (profiles/traffic.json)
  cpu_baseline — the CPU oracle (oracle/ref_cpu.py, fp32 torch CPU kernels — the reference's own CPU path) on the host cores, a
                 bounded sample extrapolated to the clip (rank 0, N = 1 only)
  config2/3    — the batch-8 action-conditioned leg: frames/s (window + context-cached), forward time, fc1 in situ, per-class ms
  config4      — BASELINE configs[4] as specified, at every N: batch 16 per GPU of 5-frame 360x640 clips with `encode_frames` of the 80 frames INSIDE the
                 timed region; forward + loss, and the whole optimisation step (bucketed RCCL gradient all-reduce when N > 1); its own roofline object
                 (dominant GEMM class of the VAE-inclusive forward, per-class fractions at M = 46 080 / 11 520) and cpu_baseline
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
T_START = time.time()       # wall clock of this process: the default run must fit the driver's window (--wall-budget)

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


# ------------------------------------------------------------------------------------------------------------------------
# launcher: one process per GPU, started before this process touches the GPU
# ------------------------------------------------------------------------------------------------------------------------
def launch_ranks(n, argv, timeout=None):
    """Starts n copies of this script as ranks 0..n-1 (env:// rendezvous on 127.0.0.1) and waits for them.  Nothing in this
    process has initialised HIP (no torch.cuda call, torch is not even imported yet), so no exec-after-GPU-init hazard.
    Returns the first non-zero exit status of a rank (the others are then terminated by PID), else 0."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    t0 = time.time()
    rc = 0
    live = list(procs)
    while live and rc == 0:
        for p in list(live):
            r = p.poll()
            if r is not None:
                live.remove(p)
                if r != 0:
                    rc = r
        if timeout is not None and time.time() - t0 > timeout:
            rc = 124
        time.sleep(0.05)
    for p in live:                       # a rank failed: stop the ranks WE started (exact PIDs), never by pattern
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-per-gpu", type=int, default=1)
    ap.add_argument("--total-frames", type=int, default=32)
    ap.add_argument("--noise-steps", type=int, default=100)
    ap.add_argument("--n-prompt", type=int, default=4)
    ap.add_argument("--use-actions", action="store_true", help="action-conditioned DiT for the headline leg")
    ap.add_argument("--algo", choices=["window", "cached", "both"], default="both",
                    help="window = recompute the whole window every noise step (reference behaviour, headline value); "
                         "cached = exact context-K/V-cached variant; both = time both (value = window)")
    ap.add_argument("--geometry", choices=["native", "g256"], default="native",
                    help="native = 360x640 frames (the reference's factories; headline); g256 = 256x256 frames, SURVEY.md 8(d) preset")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vae", action="store_true", help="skip VAE encode/decode (DiT loop only; not the headline)")
    ap.add_argument("--batched-clips", type=int, default=2,
                    help="timed clips per algorithm of the bounded batch-8 + actions leg (config2 / config3); 0 disables it")
    ap.add_argument("--batched-batch", type=int, default=8)
    ap.add_argument("--config4-steps", "--train-leg-steps", type=int, default=3, dest="config4_steps",
                    help="timed steps of each timing of the bounded config4 leg of the default run (BASELINE configs[4]: batch 16 per GPU of 5-frame 360x640 "
                         "clips, VAE encode of the 80 frames inside the timed region; forward + loss, and the whole optimisation step; every N); 0 disables it")
    ap.add_argument("--g256-clips", type=int, default=2,
                    help="timed clips of the bounded g256 leg of the default (native-geometry, N = 1) run: BASELINE.json's literal 256x256 frames through "
                         "the SURVEY.md 8(d) preset, batch 1, window algorithm; 0 disables it")
    ap.add_argument("--wall-budget", type=float, default=420.0,
                    help="seconds of wall clock the whole run may take: an optional leg (config2 / config3, config4, g256) that would not fit in front of the "
                         "CPU baseline is skipped and named in the line's `skipped_legs`; the headline leg and the CPU baseline always run")
    ap.add_argument("--cached-clips", type=int, default=3,
                    help="timed clips of the context-cached algorithm of the headline leg (--algo both; value = window): bounded, it is an extra, not the headline")
    ap.add_argument("--mode", choices=["generate", "train", "train_step"], default="generate",
                    help="train = BASELINE configs[4]: training forward + loss (train_dit.py:554-650: VAE-encode 5-frame clips, noise, "
                         "one DiT forward over the window, MSE vs the v-target), data-parallel, metric samples/s (not the headline); "
                         "train_step = the whole optimisation step (SURVEY.md 8(f)1): that forward, backward, gradient all-reduce, "
                         "clip + AdamW")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------------------------
# rank body
# ------------------------------------------------------------------------------------------------------------------------
def bench_train(args, world, rank, dev, dist, torch):
    """configs[4]: forward + loss of train_dit.py `_shared_step` (configs/train_dit_actions.yaml: batch 16 per GPU, 4 prompt
    frames + 1 target, ddim_noise_steps 50, ctx_max_noise_idx 40, clamp_min 1e-6) on synthetic 360x640 clips.  A step = one
    batch per GPU: VAE-encode 80 frames -> noise -> one DiT forward (B, T=5) -> v-target MSE; the scalar loss is all-reduced."""
    import gtav_amd.weights as W
    from gtav_amd.model.dit import DiT_models
    from gtav_amd.model.vae import VAE_models
    from gtav_amd.train import encode_frames, forward_loss
    B = args.batch_per_gpu if args.batch_per_gpu > 1 else 16
    full = args.mode == "train_step"
    dit = DiT_models["DiT-S/2"](init_weights=False, max_batch=B, trainable=full)
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

    from gtav_amd.train import training_step

    def step():
        lat = encode_frames(vae, frames)
        if full:
            loss = training_step(dit, lat, actions, tgt, ctx, ctx_noise, noise, lr=1e-5, weight_decay=0.01, max_grad_norm=1.0, world_size=world).clone()
        else:
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
    el = time.perf_counter() - t0           # this rank's time; the barrier behind it and the MAX over ranks make the figure the slowest rank's
    if world > 1:
        dist.barrier()
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = t.item()
    if rank == 0:
        flops = B * ((3 if full else 1) * dit_forward_flops(5 * P_TOK, 5, 15, 1) + 5 * 96.6e9)   # DiT forward (+ 2x for backward) + VAE encode
        extra = {}
        if full:
            applied, skipped, gnorm = dit.train_stats()
            extra = {"optimizer": "AdamW (betas 0.9 / 0.999, eps 1e-7, wd 0.01), clip_grad_norm 1.0, loss scale %g" % dit.loss_scale,
                     "last_step_applied": bool(applied), "skipped_steps": skipped, "grad_norm": gnorm}
        print(json.dumps({
            "metric": ("training step samples/sec (forward + backward + all-reduce + AdamW" if full else "training forward+loss samples/sec (") +
                      ("; 5-frame clips, configs/train_dit_actions.yaml shapes)" if full else "5-frame clips, configs/train_dit_actions.yaml shapes)"),
            **extra,
            "value": round(world * B * args.steps / el, 3), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(el / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp16 (fp32 accumulate/residual)", "data": "synthetic", "loss": float(loss.item()),
            "config": {"workload": ("SURVEY.md 8(f)1: train_dit.py optimisation step" if full else "BASELINE configs[4]: train_dit.py forward+loss") +
                                   ", batch %d per GPU, DiT-S/2 + VAE encode of %d frames" % (B, 5 * B),
                       "global_batch": world * B,
                       "parallelism": ("data-parallel x%d (one all-reduce of the 2.4 GB fp32 gradient arena per step)" if full else
                                       "data-parallel x%d (forward only; loss all-reduce)") % world},
            "achieved_tflops_per_gpu": round(flops * args.steps / el / 1e12, 1)}))


def bench_config4(args, world, rank, dev, dist, torch):
    """The config4 object of the default line (see the call site).  Roofline object: the GEMM class with the largest time per VAE-inclusive forward, among the
    VAE encoder's four (M = 80 x 576 = 46 080 tokens, one pass) and the DiT's four (M = 16 x 5 x 144 = 11 520), timed in situ with dispatch-attached HIP
    events (gtav_vae_profile / gtav_dit_profile); algorithmic FLOPs per launch 2 M N K."""
    import gtav_amd.weights as W
    from gtav_amd.model.dit import DiT_models
    from gtav_amd.model.vae import VAE_models
    from gtav_amd.train import encode_frames, forward_loss, training_step
    TB, NF = 16, 80
    n = args.config4_steps
    tdit = DiT_models["DiT-S/2"](init_weights=False, max_batch=TB, trainable=True)
    tdit.load_state_dict(W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0))
    vae4 = VAE_models["vit-l-20-shallow-encoder"](init_weights=False, max_frames_per_call=NF)
    vae4.load_state_dict(W.synth_state_dict(W.vae_param_shapes(), seed=1))
    g = torch.Generator().manual_seed(100 + rank)
    frames = torch.rand(TB, 5, 3, 360, 640, generator=g).to(dev)
    tact = torch.zeros(TB, 5, 25, device=dev)
    tact[:, :, 3] = 1
    tgt, ctx = torch.randint(1, 51, (TB,), generator=g), torch.randint(1, 41, (TB,), generator=g)
    cn, nz = torch.randn(TB, 4, 16, 18, 32, generator=g).to(dev), torch.randn(TB, 1, 16, 18, 32, generator=g).to(dev)

    def fwd():
        lat = encode_frames(vae4, frames)
        loss, _, _ = forward_loss(tdit, lat, tact, tgt, ctx, cn, nz)
        if world > 1:
            loss = loss.clone()
            dist.all_reduce(loss, op=dist.ReduceOp.SUM)
            loss /= world
        return loss

    def step():
        lat = encode_frames(vae4, frames)
        return training_step(tdit, lat, tact, tgt, ctx, cn, nz, lr=1e-5, weight_decay=0.01, max_grad_norm=1.0, world_size=world)

    def enc():
        return encode_frames(vae4, frames)

    def timed(fn, k):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            out = fn()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0       # (taken before the closing barrier: no collective inside the timing; the MAX over ranks is the slowest rank)
        if world > 1:
            dist.barrier()
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        return el / k, out

    fwd()
    step()                                   # warm-up of both paths (workspaces, RCCL buckets)
    t_fwd, loss_f = timed(fwd, n)
    t_step, loss_s = timed(step, n)
    t_enc, _ = timed(enc, n)
    applied, skipped, gnorm = tdit.train_stats()
    # ---- N > 1: one more step with an event pair around every bucket's all-reduce on the communication stream (outside the timed regions): the first SCALE run
    # then shows which buckets hide behind the backward pass and what the ring reaches per bucket ----
    buckets_report = None
    if world > 1:
        bt = []
        training_step(tdit, enc(), tact, tgt, ctx, cn, nz, lr=1e-5, weight_decay=0.01, max_grad_norm=1.0, world_size=world, bucket_timings=bt)
        torch.cuda.synchronize()
        per = [{"after_phase": int(ph), "mbytes": round(nb / 1e6, 1), "ms": round(e0.elapsed_time(e1), 3),
                "bus_gbps": round(2.0 * (world - 1) / world * nb / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1) if e0.elapsed_time(e1) > 0 else None} for ph, nb, e0, e1 in bt]
        buckets_report = {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(), "buckets": per,
                          "sum_ms": round(sum(b["ms"] for b in per), 2), "total_mbytes": round(sum(b["mbytes"] for b in per), 1),
                          "note": "in-situ: each all-reduce runs on the communication stream beside the backward pass of the blocks still being differentiated; "
                                  "bus_gbps = 2 (N - 1) / N x bytes / time, the ring's per-link figure (xGMI: ~153 GB/s per link)"}
    fl_dit = TB * dit_forward_flops(5 * P_TOK, 5, 15, 1)
    fl_vae = NF * GEOM["native"]["vae_gflop"][0] * 1e9
    # ---- in-situ per-class kernel times: two profiled encodes + two profiled forwards (outside the timed regions) ----
    vae4.profile(True)
    for _ in range(2):
        enc()
    vprof = debias(vae4.profile_read())
    vae4.profile(False)
    tdit.profile(True)
    for _ in range(2):
        forward_loss(tdit, enc(), tact, tgt, ctx, cn, nz)
    dprof = debias(tdit.profile_read())
    tdit.profile(False)
    Mv, Md = NF * 576, TB * 5 * P_TOK
    gf = {"vae_qkv": (vprof["gemm_qkv"], 2.0 * Mv * 3072 * 1024, "VAE qkv GEMM + bias / partial-RoPE / head-layout epilogue (N=3072 K=1024)"),
          "vae_proj": (vprof["gemm_proj"], 2.0 * Mv * 1024 * 1024, "VAE attention projection, in-place residual epilogue (N=1024 K=1024)"),
          "vae_fc1": (vprof["gemm_fc1"], 2.0 * Mv * 4096 * 1024, "VAE fc1 GEMM + erf-GELU epilogue (N=4096 K=1024)"),
          "vae_fc2": (vprof["gemm_fc2"], 2.0 * Mv * 1024 * 4096, "VAE fc2 GEMM, in-place residual epilogue (N=1024 K=4096)"),
          "dit_qkv": (dprof["gemm_qkv"], 2.0 * Md * 3072 * 1024, "DiT to_qkv GEMM + RoPE / head-layout epilogue (N=3072 K=1024)"),
          "dit_out": (dprof["gemm_out"], 2.0 * Md * 1024 * 1024, "DiT out-proj GEMM (N=1024 K=1024)"),
          "dit_fc1": (dprof["gemm_fc1"], 2.0 * Md * 4096 * 1024, "DiT fc1 GEMM + GELU-tanh epilogue (N=4096 K=1024)"),
          "dit_fc2": (dprof["gemm_fc2"], 2.0 * Md * 1024 * 4096, "DiT fc2 GEMM (N=1024 K=4096)")}
    per_class, tot_fl, tot_ms = {}, 0.0, 0.0
    for k, ((ms, cnt), fl, _) in gf.items():
        if cnt:
            us = ms / cnt * 1e3
            per_class[k] = {"us_per_launch": round(us, 2), "launches_per_forward": int(cnt // 2), "ms_per_forward": round(ms / 2, 3),
                            "frac_of_mfma_peak": round(fl / (us * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
            tot_fl += fl * cnt
            tot_ms += ms
    att_ms, att_n = vprof["attn_spatial"]
    att_fl = 4.0 * 576 * 576 * 64 * 16 * NF
    other = {"vae_attn_spatial_S576": {"us_per_launch": round(att_ms / max(att_n, 1) * 1e3, 2), "launches_per_forward": int(att_n // 2),
                                       "frac_of_mfma_peak": round(att_fl / (att_ms / max(att_n, 1) * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4) if att_n else None},
             "vae_ln_affine": {"us_per_launch": round(vprof["ln_affine"][0] / max(vprof["ln_affine"][1], 1) * 1e3, 2), "launches_per_forward": int(vprof["ln_affine"][1] // 2)},
             "dit_ln_modulate": {"us_per_launch": round(dprof["ln_modulate"][0] / max(dprof["ln_modulate"][1], 1) * 1e3, 2), "launches_per_forward": int(dprof["ln_modulate"][1] // 2)},
             "vae_patchify_embed_quant_ms_per_forward": round(vprof["other"][0] / 2, 3)}
    if not per_class:
        raise RuntimeError("bench_config4: the profiler recorded no GEMM launch (gtav_vae_profile / gtav_dit_profile returned empty classes)")
    dom = max(per_class, key=lambda k: per_class[k]["ms_per_forward"])
    (ms_d, n_d), fl_d, desc = gf[dom]
    ach = fl_d / (ms_d / n_d * 1e-3) / 1e12
    # HBM bytes per launch of that class from the committed rocprofv3 --pmc passes (profiles/traffic.json: the DiT's classes at M = 11 520, the VAE's at M = 46 080)
    traffic, mfma_busy, tsrc = traffic_for(dom if dom.startswith("vae") else dom.replace("dit_", ""), Mv if dom.startswith("vae") else Md)
    roofline = {"kernel": "%s, csrc/gemm.hip, M=%d, fp16 MFMA (the GEMM class with the largest time per VAE-inclusive forward)" % (desc, Mv if dom.startswith("vae") else Md),
                "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "mfma_busy_pmc": mfma_busy, "traffic_source": tsrc,
                "avg_launch_us": round(ms_d / n_d * 1e3, 2), "launches_timed": int(n_d), "flops_per_launch": fl_d,
                "timing": "HIP events attached to the dispatch (hipExtLaunchKernel), less the calibrated offset of the pair (timer.bias_us of the line)",
                "gemm_aggregate_frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4) if tot_ms > 0 else None,
                "per_class": per_class, "other_classes": other}
    out = {"workload": "BASELINE configs[4] as specified: train_dit.py forward + loss (configs/train_dit_actions.yaml: batch %d per GPU x %d GPU(s), 5-frame 360x640 clips, "
                       "ddim_noise_steps 50, ctx_max_noise_idx 40, clamp_min 1e-6), encode_frames of the %d frames per GPU INSIDE the timed region; second timing: the "
                       "whole optimisation step (that forward, backward, gradient all-reduce, clip_grad_norm 1.0, AdamW)" % (TB, world, NF),
           "steps_timed": n, "world_size": world,
           "forward_loss": {"ms_per_step": round(t_fwd * 1e3, 2), "samples_per_s": round(world * TB / t_fwd, 2), "loss": float(loss_f),
                            "executed_tflop_per_gpu_step": round((fl_dit + fl_vae) / 1e12, 3), "achieved_tflops_per_gpu": round((fl_dit + fl_vae) / t_fwd / 1e12, 1),
                            "frac_of_mfma_peak": round((fl_dit + fl_vae) / t_fwd / 1e12 / MFMA_PEAK_TFLOPS, 4)},
           "train_step": {"ms_per_step": round(t_step * 1e3, 2), "samples_per_s": round(world * TB / t_step, 2), "loss": float(loss_s),
                          "achieved_tflops_per_gpu": round((3 * fl_dit + fl_vae) / t_step / 1e12, 1), "last_step_applied": bool(applied), "skipped_steps": skipped,
                          "grad_norm": round(gnorm, 4),
                          "gradient_all_reduce": ("RCCL, world size %d: bucketed all-reduce of the 2.4 GB fp32 gradient arena, overlapped with the backward pass" % world)
                          if world > 1 else "none (one rank)",
                          "gradient_all_reduce_buckets": buckets_report,
                          "dtype": "fp16 operands, fp32 accumulate / master weights / gradients / AdamW state, loss scale %g" % tdit.loss_scale},
           "vae_encode_80_frames": {"ms": round(t_enc * 1e3, 3), "tflops": round(fl_vae / t_enc / 1e12, 1), "frac_of_mfma_peak": round(fl_vae / t_enc / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                    "frames_per_call": NF},
           "roofline": roofline}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the reference's CPU path for the same step, ONE sample: VAE encode of its 5 frames + DiT forward (B = 1, T = 5) + loss; and autograd + AdamW for the step
        from oracle import ref_cpu as O
        cores = host_cores()
        torch.set_num_threads(cores)
        sd, vsd = tdit.state_dict(), vae4.state_dict()
        with torch.no_grad():
            t0 = time.perf_counter()
            lat1 = O.vae_encode_frames(vsd, O.vit_l_20_shallow_encoder(), frames[:1].cpu())
            t_e = time.perf_counter() - t0
            t1 = torch.tensor([[19, 19, 19, 19, 500]])
            t0 = time.perf_counter()
            O.dit_forward(sd, O.dit_s_2(), lat1, t1, tact[:1].cpu())
            t_f = time.perf_counter() - t0
        t0 = time.perf_counter()
        _, _, grads = O.dit_loss_and_grads(sd, O.dit_s_2(), lat1, t1, tact[:1].cpu(), nz[:1].cpu())
        t_b = time.perf_counter() - t0
        params = {k: v for k, v in sd.items() if not k.endswith("freqs")}
        t0 = time.perf_counter()
        O.adamw_reference(params, grads, lr=1e-5, weight_decay=0.01, max_grad_norm=1.0, steps=1)
        t_o = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(1.0 / (t_e + t_f), 4), "unit": "samples/s (forward + loss)", "cores": cores, "kind": "port",
                               "train_step_samples_per_s": round(1.0 / (t_e + t_b + t_o), 4),
                               "sample": "oracle/ref_cpu.py (fp32 torch CPU), ONE 5-frame sample: VAE encode of its 5 frames %.2f s + DiT forward %.2f s; for the step: "
                                         "autograd forward + backward %.2f s + clip_grad_norm_ + AdamW over the 608 M parameters %.2f s" % (t_e, t_f, t_b, t_o)}
        del grads, params, sd, vsd
    del tdit, vae4, frames, cn, nz
    torch.cuda.empty_cache()
    return out


def rank_main(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    import torch
    have = torch.cuda.device_count()          # does not initialise the GPU
    # Rehearsal of the N > 1 control flow on a ONE-GPU box (tools/bench_rehearsal.sh): every rank uses cuda:0 and the collectives run over gloo.  Its numbers mean
    # nothing (the ranks share one GPU) and the line says so; RCCL itself refuses two ranks on one device.
    rehearse = os.environ.get("GTAV_BENCH_REHEARSE_ONE_GPU") == "1" and world > 1
    if rehearse:
        local_rank = 0
    if have < (1 if rehearse else world) or not torch.cuda.is_available():
        sys.stderr.write(f"bench.py rank {rank}/{world}: needs {world} GPUs, this host has {have}; there is no CPU fallback for the product path\n")
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
        assert dist.get_world_size() == args.gpus
    try:
        if args.mode in ("train", "train_step"):
            bench_train(args, world, rank, dev, dist, torch)
        else:
            bench_generate(args, world, rank, dev, dist, torch)
    finally:
        if world > 1:
            dist.destroy_process_group()


def traffic_for(cls, M):
    """HBM bytes per launch and MFMA-busy fraction of GEMM class `cls` ("qkv", "out", "fc1", "fc2") at M tokens from the committed rocprofv3
    --pmc passes (profiles/traffic.json, written by tools/gemm_traffic.sh from the torch-free driver tools/gemm_pmc).  The file records the sha
    of csrc/gemm.hip it was measured on: a different kernel source means the numbers are stale and they are reported as null instead."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, None, "profiles/traffic.json missing"
    tj = json.load(open(tpath))
    sha = hashlib.sha256(open(os.path.join(ROOT, "ai-generated-gtav_amd", "csrc", "gemm.hip"), "rb").read()).hexdigest()[:16]
    if tj.get("gemm_hip_sha16") != sha:
        return None, None, "profiles/traffic.json was measured on another build of csrc/gemm.hip (stale)"
    ent = tj.get("%s_M%d" % (cls, M)) or {}
    return ent.get("hbm_bytes_per_launch"), ent.get("mfma_busy"), tj.get("source", "")


_TIMER = None


def timer_calibration():
    """Constant offset of the in-situ profiler's readings (HIP events attached to a kernel's dispatch): a one-wave kernel spins a known time on the device's own
    100 MHz clock under the same attached event pairs (gtav_timer_calibrate, 5 us and 20 us spins, 64 back-to-back launches each): `event_minus_device_us` is what
    the pair reads beyond the kernel's own figure.  rocprofv3's kernel duration also contains the dispatch ramp (dispatch -> first instruction, last instruction
    -> completion signal): `rocprof_minus_device_us`, measured by rocprofv3 on the same spin kernel (profiles/timer_calibration.json, tools/timer_calibration.sh).
    bias_us = event_minus_device_us - rocprof_minus_device_us is subtracted from every per-launch time, which puts them on rocprofv3's scale."""
    global _TIMER
    if _TIMER is not None:
        return _TIMER
    import ctypes as C
    from gtav_amd import lib as L
    import torch
    out = {}
    for spin in (5, 20):
        e, d = C.c_double(0), C.c_double(0)
        L.check(L.load().gtav_timer_calibrate(spin, 64, C.byref(e), C.byref(d), torch.cuda.current_stream().cuda_stream))
        out[spin] = (e.value, d.value)
    off = sum(e - d for e, d in out.values()) / len(out)
    ramp, src = None, "profiles/timer_calibration.json missing: readings are NOT corrected"
    tpath = os.path.join(ROOT, "profiles", "timer_calibration.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        ramp, src = tj.get("rocprof_minus_device_us"), tj.get("source", "profiles/timer_calibration.json")
    bias = max(0.0, off - ramp) if ramp is not None else 0.0
    _TIMER = {"bias_us": round(bias, 3), "event_minus_device_us": round(off, 3), "rocprof_minus_device_us": ramp,
              "spin_5us": {"event_us": round(out[5][0], 3), "device_us": round(out[5][1], 3)},
              "spin_20us": {"event_us": round(out[20][0], 3), "device_us": round(out[20][1], 3)},
              "method": "every us_per_launch / frac in this line = (event-pair reading - bias_us): the offset of an event pair attached to a dispatch, calibrated on a "
                        "kernel of known device duration, less the dispatch ramp rocprofv3 counts too", "ramp_source": src}
    return _TIMER


def debias(prof):
    """Profiler readings {class: (ms summed over n launches, n)} on rocprofv3's scale: the single-kernel classes (events attached to the dispatch) lose
    timer_calibration()'s bias per launch; "other" and the empty pair are plain event records around several launches and stay as read."""
    b = timer_calibration()["bias_us"] * 1e-3
    return {k: ((max(ms - n * b, 0.0), n) if k not in ("other", "empty_event_pair") else (ms, n)) for k, (ms, n) in prof.items()}


def bench_generate(args, world, rank, dev, dist, torch):
    global P_TOK
    import gtav_amd.weights as W
    from gtav_amd.generate import generate_clip, shard_inputs
    from gtav_amd.model.dit import DiT, DiT_models
    from gtav_amd.model.vae import AutoencoderKL, VAE_models

    B = args.batch_per_gpu
    Bb = args.batched_batch if (args.batched_clips > 0 and args.geometry == "native") else 0   # batch of the config2 / config3 leg
    total, n_prompt, steps = args.total_frames, args.n_prompt, args.noise_steps
    geo = GEOM[args.geometry]
    (FH, FW), (LH, LW) = geo["frame"], geo["lat"]
    P_TOK = (LH // 2) * (LW // 2)
    # ---- models with deterministic synthetic weights (every matrix non-zero, incl. adaLN) ----
    Bmax = max(B, Bb)
    if args.geometry == "native":
        dit = DiT_models["DiT-S/2"](init_weights=False, max_batch=Bmax)
        dit.load_state_dict(W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0))
    else:
        dit = DiT(**geo["dit"], init_weights=False, max_batch=Bmax)
        dit.load_state_dict(W.synth_state_dict(W.dit_param_shapes(**geo["dit"]), seed=0))
    dit.reserve(Bmax, 5, steps)      # one workspace for both legs: nothing is rebuilt inside a timed region
    vae = None
    if not args.no_vae:
        nfc = min(128, 32 * Bmax)        # frames per VAE call (workspace cap): the 32 frames of a clip in one decode at batch 1, 128-frame chunks at batch 8
        if args.geometry == "native":
            vae = VAE_models["vit-l-20-shallow-encoder"](init_weights=False, max_frames_per_call=nfc)
            vae.load_state_dict(W.synth_state_dict(W.vae_param_shapes(), seed=1))
        else:
            vae = AutoencoderKL(**geo["vae"], init_weights=False, max_frames_per_call=nfc)
            vae.load_state_dict(W.synth_state_dict(W.vae_param_shapes(**geo["vae"]), seed=1))

    def leg_inputs(b, use_actions, seed):
        # synthetic inputs are a function of the GLOBAL sample id: any sharding of the batch sees the same sequences
        _, frames, noise = shard_inputs(world * b, rank, world, n_prompt, total, (FH, FW), (LH, LW), seed=seed)
        actions = None
        if use_actions:
            actions = torch.zeros(b, total, 25, device=dev)
            actions[:, :, 3] = 1  # "W" for every frame (generate.py:159,181)
        return frames.to(dev), noise.to(dev), actions

    def one_clip(inp, cached, nsteps):
        frames, noise, actions = inp
        if vae is None:
            from gtav_amd.generate import all_gather_latents, generate_latents
            lat = torch.randn(frames.shape[0], n_prompt, 16, LH, LW, generator=torch.Generator().manual_seed(7)).to(dev) * 0.5
            return all_gather_latents(generate_latents(dit, lat, total, nsteps, noise, actions, ctx_cache=cached)), None
        return generate_clip(dit, vae, frames, noise, total, nsteps, actions, ctx_cache=cached)

    def timed(inp, cached, nclips, nwarm, short_warm=False):
        for _ in range(nwarm):
            one_clip(inp, cached, 2 if short_warm else steps)   # short warm-up: same shapes / graph keys, 3 forwards per frame
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nclips):
            one_clip(inp, cached, steps)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0       # (taken before the closing barrier: no collective inside the timing; the MAX over ranks is the slowest rank)
        if world > 1:
            dist.barrier()
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        return el

    def fused_gflop(M, frames_b, frames_t, prof, classes, gflop):
        """The fused to_qkv + attention launches (gtav_dit_fused_launches) are booked under the attention class of their half — one class, one kernel.  Their algorithmic
        FLOPs per launch are the to_qkv GEMM's 2 M N K plus that attention's; they join `gflop` so that the GEMM aggregate and the choice of the dominant kernel see them."""
        mask = dit.fused_launches(frames_b, frames_t)
        fl_qkv = 2.0 * M * 3 * D_MODEL * D_MODEL
        if mask & 1 and prof["attn_spatial"][1]:
            gflop["attn_spatial"] = fl_qkv + 4.0 * (M // P_TOK) * (D_MODEL // 64) * P_TOK * P_TOK * 64
            classes["attn_spatial"]["kernel"] = "fused spatial to_qkv GEMM + RoPE + attention launch (csrc/gemm.hip gemm_qkvs_attn_kernel): FLOPs = 2 M N K + 4 frames heads P^2 64"
            tr_b, tr_busy, _ = traffic_for("qkvs", M)
            if tr_b:
                classes["attn_spatial"]["hbm_bytes_per_launch_pmc"] = tr_b
                classes["attn_spatial"]["algorithmic_bytes_per_launch"] = int(2.0 * (3 * D_MODEL * D_MODEL + 2 * M * D_MODEL))   # W + X + attention output, fp16
                classes["attn_spatial"]["mfma_busy_pmc"] = tr_busy
        if mask & 2 and prof["attn_temporal"][1]:
            gflop["attn_temporal"] = fl_qkv + 4.0 * (M // frames_t) * 64 * (D_MODEL // 64) * (frames_t + 1) / 2
            classes["attn_temporal"]["kernel"] = "fused temporal to_qkv GEMM + RoPE + causal attention launch (csrc/gemm.hip gemm_qkvt_attn_kernel)"
        return mask

    def profile_forward(b, actions):
        """in-situ per-class kernel times over real forwards of (b, T = 5): dispatch-attached HIP events (gtav_dit_profile)"""
        g = torch.Generator().manual_seed(3)
        xw = torch.randn(b, 5, 16, LH, LW, generator=g).to(dev)
        tw = torch.tensor([[15, 15, 15, 15, 500]] * b)
        aw = actions[:, :5].contiguous() if actions is not None else None
        for _ in range(2):
            dit(xw, tw, aw)
        torch.cuda.synchronize()
        dit.profile(True)
        nprof = 6
        for _ in range(nprof):
            dit(xw, tw, aw)
        torch.cuda.synchronize()
        prof = debias(dit.profile_read())
        dit.profile(False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nfw = 10
        for _ in range(nfw):
            dit(xw, tw, aw)
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - t0) / nfw * 1e3
        M = b * 5 * P_TOK
        flops_fc1 = 2.0 * M * HM * D_MODEL
        ev_ms, ev_n = prof.pop("empty_event_pair")
        classes = {k: {"ms_per_forward": round(v[0] / nprof, 4), "launches_per_forward": v[1] // nprof} for k, v in prof.items()}
        # per GEMM class: achieved TFLOP/s from its algorithmic FLOPs (2 M N K per launch)
        gflop = {"gemm_qkv": 2.0 * M * 3 * D_MODEL * D_MODEL, "gemm_out": 2.0 * M * D_MODEL * D_MODEL, "gemm_fc1": flops_fc1,
                 "gemm_fc2": flops_fc1}
        fused_mask = fused_gflop(M, b, 5, prof, classes, gflop)
        shape_of = {"gemm_qkv": "to_qkv GEMM + RoPE / head-layout epilogue (N=3072 K=1024)", "gemm_out": "out-proj GEMM (N=1024 K=1024, residual epilogue / split-K slabs)",
                    "gemm_fc1": "fc1 GEMM + GELU-tanh epilogue (N=4096 K=1024)", "gemm_fc2": "fc2 GEMM (N=1024 K=4096, residual epilogue / split-K slabs)",
                    "attn_spatial": "fused spatial to_qkv GEMM + RoPE + attention launch (N=3072 K=1024; gemm_qkvs_attn_kernel)",
                    "attn_temporal": "fused temporal to_qkv GEMM + RoPE + attention launch (N=3072 K=1024; gemm_qkvt_attn_kernel)"}
        tot_fl = tot_ms = 0.0
        for k, fl in gflop.items():
            ms, n = prof[k]
            if n:
                classes[k]["us_per_launch"] = round(ms / n * 1e3, 2)
                classes[k]["tflops"] = round(fl / (ms / n * 1e-3) / 1e12, 1)
                classes[k]["frac_of_mfma_peak"] = round(fl / (ms / n * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)
                tot_fl += fl * n
                tot_ms += ms
        # the roofline object describes the DOMINANT kernel: the GEMM class with the largest time per forward (not the best-looking one)
        dom = max(gflop, key=lambda k: prof[k][0])
        ms_d, n_d = prof[dom]
        avg_ms = ms_d / max(n_d, 1)
        ach = gflop[dom] / (avg_ms * 1e-3) / 1e12
        traffic, mfma_busy, tsrc = traffic_for(dom.replace("gemm_", ""), M)
        worst = min(gflop, key=lambda k: classes[k].get("tflops", 1e30))
        roofline = {"kernel": "%s, csrc/gemm.hip, M=%d, fp16 MFMA (the GEMM class with the largest time per forward)" % (shape_of[dom], M), "bound": "mfma",
                    "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                    "traffic": traffic, "mfma_busy_pmc": mfma_busy, "traffic_source": tsrc, "avg_launch_us": round(avg_ms * 1e3, 2), "launches_timed": int(n_d),
                    "flops_per_launch": gflop[dom],
                    "timing": "HIP events attached to the dispatch (hipExtLaunchKernel), less the calibrated offset of the pair (timer.bias_us of the line)",
                    "timer_bias_us": timer_calibration()["bias_us"], "empty_event_pair_us": round(ev_ms / max(ev_n, 1) * 1e3, 2),
                    "gemm_aggregate_frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4) if tot_ms > 0 else None,
                    "gemm_aggregate_note": "FLOP-weighted over the GEMM classes (the four plain ones and, where they run, the fused to_qkv + attention launches with their attention FLOPs): "
                                           "sum of FLOPs over their launches / sum of their launch times",
                    "fused_launches": {"spatial_to_qkv_attention": bool(fused_mask & 1), "temporal_to_qkv_attention": bool(fused_mask & 2)},
                    "worst_class": {"kernel": shape_of[worst], "frac": classes[worst].get("frac_of_mfma_peak")},
                    "per_class_frac": {k.replace("gemm_", ""): classes[k].get("frac_of_mfma_peak") for k in gflop}}
        fwd_flops = dit_forward_flops(M, b * 5, 15, b)
        step_tflops = fwd_flops / (fwd_ms * 1e-3) / 1e12
        dit_step = {"forward_ms_B%d_T5" % b: round(fwd_ms, 3), "executed_tflop_per_forward": round(fwd_flops / 1e12, 4),
                    "achieved_tflops": round(step_tflops, 1), "frac_of_mfma_peak": round(step_tflops / MFMA_PEAK_TFLOPS, 4),
                    "kernel_classes": classes}
        return roofline, dit_step, (xw, tw)

    def profile_cached_step(b, actions):
        """in-situ per-class kernel times of the CONTEXT-CACHED sampler step of batch b (M = b x P tokens: the frame being denoised only, context K / V
        from the temporal caches; generate.py:200-220 with the exact caching of DESIGN.md 5) + the captured-graph time of that step.  The roofline
        object follows SURVEY.md 8(d)'s accounting rule: FLOPs actually executed by this algorithm's step."""
        from gtav_amd.utils import alphas_cumprod
        nst = 24
        g = torch.Generator().manual_seed(5)
        F = 6
        xb = (torch.randn(b, F, 16, LH, LW, generator=g) * 0.5).to(dev)
        ab = actions[:, :F].contiguous() if actions is not None else None
        dit.set_schedule(alphas_cumprod(1e-4))
        ts = [999 - 9 * k for k in range(nst + 1)]

        xx = torch.empty_like(xb)         # ONE buffer: the captured graph is keyed by it

        def frame(profile):
            xx.copy_(xb)
            dit.prepare_frame_(b, F, 1, 5, 15, ts, ab)
            dit.denoise_step_(xx, 1, 5, 15, ts[0], ts[1], False, ab, cached=False, cond_step=0)
            torch.cuda.synchronize()
            if profile:
                dit.profile(True)
            t0 = time.perf_counter()
            for k in range(1, nst + 1):
                dit.denoise_step_(xx, 1, 5, 15, ts[k], ts[min(k + 1, nst)], False, ab, cached=True, cond_step=k)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / nst * 1e3

        frame(False)                      # eager warm-up + capture of this key
        graph_ms = min(frame(False), frame(False))
        frame(True)
        prof = debias(dit.profile_read())
        dit.profile(False)
        M = b * P_TOK
        ev_ms, ev_n = prof.pop("empty_event_pair")
        classes = {k: {"ms_per_step": round(v[0] / nst, 4), "launches_per_step": v[1] // nst} for k, v in prof.items()}
        gflop = {"gemm_qkv": 2.0 * M * 3 * D_MODEL * D_MODEL, "gemm_out": 2.0 * M * D_MODEL * D_MODEL, "gemm_fc1": 2.0 * M * HM * D_MODEL,
                 "gemm_fc2": 2.0 * M * HM * D_MODEL}
        fused_gflop(M, b, 1, prof, classes, gflop)
        tot_fl = tot_ms = 0.0
        for k, fl in gflop.items():
            ms, n = prof[k]
            if n:
                classes[k]["us_per_launch"] = round(ms / n * 1e3, 2)
                classes[k]["frac_of_mfma_peak"] = round(fl / (ms / n * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)
                tot_fl += fl * n
                tot_ms += ms
        step_fl = dit_forward_flops(M, b, 5, b)
        w_bytes = 2.0 * (DEPTH * 2 * 12 * D_MODEL * D_MODEL)          # fp16 GEMM weights streamed once per step (0.40 GB at DiT-S/2) ...
        w_bytes_all = 1.216e9                                          # ... SURVEY.md 8(d)'s per-forward figure (all 607.9 M parameters as 2-byte values)
        mfma = step_fl / (graph_ms * 1e-3) / 1e12
        hbm = w_bytes_all / (graph_ms * 1e-3) / 1e9
        bound = "hbm" if b * P_TOK < 310 else "mfma"                   # ridge at ~310 FLOP/B = ~310 tokens per weight byte pair
        roof = {"kernel": "context-cached sampler step (one hipGraph replay: %d launches), M=%d tokens" % (sum(v[1] for v in prof.values()) // nst, M),
                "bound": bound,
                "achieved": round(hbm if bound == "hbm" else mfma, 2), "peak": 8000.0 if bound == "hbm" else MFMA_PEAK_TFLOPS,
                "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
                "frac": round((hbm / 8000.0) if bound == "hbm" else (mfma / MFMA_PEAK_TFLOPS), 4), "traffic": None,
                "algorithmic_bytes_per_step": w_bytes_all, "fp16_gemm_weight_bytes_per_step": w_bytes, "executed_tflop_per_step": round(step_fl / 1e12, 4),
                "mfma_tflops": round(mfma, 1), "weight_stream_gbps": round(hbm, 1),
                "gemm_aggregate_frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4) if tot_ms > 0 else None,
                "note": "whole-step figure (executed FLOPs / weight bytes of ONE cached step over the captured step's time), not one kernel: at this size "
                        "the step is a chain of ~230 short launches"}
        return roof, {"graph_ms_per_step": round(graph_ms, 4), "tokens": M, "kernel_classes": classes,
                      "empty_event_pair_us": round(ev_ms / max(ev_n, 1) * 1e3, 2)}

    def algo_report(b, results):
        """results: algo -> (seconds, clips timed)"""
        out = {}
        for algo, (e, nclips) in results.items():
            fl = clip_flops(b, n_prompt, total, steps, 5, algo == "cached") * world
            out["algo_" + algo] = {"generated_frames_per_s": round(world * b * (total - n_prompt) * nclips / e, 4), "clips_timed": nclips,
                                   "ms_per_clip": round(e / nclips * 1e3, 1), "executed_pflop_per_clip": round(fl / 1e15, 4),
                                   "achieved_tflops_per_gpu": round(fl / world * nclips / e / 1e12, 1)}
        return out

    # ---- wall-clock accounting: every leg's wall time goes into the line; an OPTIONAL leg is skipped when it would push the run past --wall-budget ----
    leg_wall, skipped_legs = {}, []
    t_leg = [time.time()]

    def leg_done(name):
        now = time.time()
        leg_wall[name] = round(now - t_leg[0], 1)
        t_leg[0] = now

    CPU_RESERVE = 50.0 if (rank == 0 and world == 1 and not args.no_cpu_baseline) else 5.0

    def fits(name, estimate_s):
        """rank 0 decides (its clock), every rank follows: an optional leg runs only if it is expected to end before the CPU baseline's reserve"""
        ok = (time.time() - T_START) + estimate_s + CPU_RESERVE <= args.wall_budget
        if world > 1:
            t = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
            dist.broadcast(t, src=0)
            ok = bool(t.item())
        if not ok:
            skipped_legs.append({"leg": name, "estimate_s": estimate_s, "elapsed_s": round(time.time() - T_START, 1)})
        return ok

    # ---- the next-weight L2 prefetch pays on some boxes and not on others (bit-identical results either way): time both on this box, keep the faster ----
    prefetch_choice = None
    if 256 <= B * 5 * P_TOK <= 1536:
        from gtav_amd.generate import tune_weight_prefetch
        prefetch_choice = tune_weight_prefetch(dit, B, use_actions=args.use_actions, latent_hw=(LH, LW))
    # ---- headline leg: batch B per GPU (configs[1] by default) ----
    leg_done("setup_models_and_prefetch_tuner")
    inp = leg_inputs(B, args.use_actions, seed=1000)
    algos = ["window", "cached"] if args.algo == "both" else [args.algo]
    head = "window" if "window" in algos else algos[0]
    results = {}
    for algo in algos:
        if algo == head:         # the headline: exactly --steps timed clips after --warmup untimed ones
            results[algo] = (timed(inp, algo == "cached", args.steps, args.warmup), args.steps)
        else:                    # the extra algorithm: bounded (--cached-clips), one short warm-up clip with the same shapes / graph keys
            nc = max(1, min(args.steps, args.cached_clips))
            results[algo] = (timed(inp, algo == "cached", nc, 1, short_warm=True), nc)
        leg_done("headline_" + algo)
    el = results[head][0]
    gen_frames = world * B * (total - n_prompt)
    value = gen_frames * args.steps / el
    roofline, dit_step, (xw, tw) = profile_forward(B, inp[2])
    roofline_cached = dit_step_cached = None
    if "cached" in results:
        roofline_cached, dit_step_cached = profile_cached_step(B, inp[2])
    leg_done("headline_in_situ_profiles")

    # ---- batched leg: batch 8 per GPU, action-conditioned (configs[2] at N = 1, configs[3]-shaped at N > 1), bounded ----
    batched = None
    # (estimate: per algorithm one 2-step warm-up clip + --batched-clips clips of ~20 s window / ~8 s cached at batch 8, + profiles)
    if Bb > 0 and Bb != B and fits("config2" if world == 1 else "config3", (args.batched_clips * 28.0 + 12.0) * Bb / 8.0):
        binp = leg_inputs(Bb, True, seed=5000)
        bpf = None
        if "cached" in algos and 256 <= Bb * P_TOK <= 1536:      # the context-cached step of this leg (Bb x one frame of tokens) is in the prefetch's range: tune it at that size
            from gtav_amd.generate import tune_weight_prefetch
            bpf = tune_weight_prefetch(dit, Bb, window=1, use_actions=True, latent_hw=(LH, LW))
        bres = {algo: (timed(binp, algo == "cached", args.batched_clips, 1, short_warm=True), args.batched_clips) for algo in algos}
        broof, bstep, _ = profile_forward(Bb, binp[2])
        broof_c, bstep_c = profile_cached_step(Bb, binp[2]) if "cached" in bres else (None, None)
        bel = bres["window" if "window" in bres else algos[0]][0]
        batched = {"workload": "BASELINE configs[%d]: batch %d per GPU x %d GPU(s) = %d sequences, action-conditioned, %d frames (%d prompt), "
                               "%d noise steps, VAE inside the timed region" % (2 if world == 1 else 3, Bb, world, world * Bb, total, n_prompt, steps),
                   "clips_timed": args.batched_clips, "warmup": "1 clip with 2 noise steps (same shapes and hipGraph keys)",
                   "value": round(world * Bb * (total - n_prompt) * args.batched_clips / bel, 4), "unit": "generated frames/s",
                   "roofline": broof, "dit_step": bstep, "roofline_cached": broof_c, "dit_step_cached": bstep_c, "weight_prefetch_cached_step": bpf}
        batched.update(algo_report(Bb, bres))
        del binp
        leg_done("config2" if world == 1 else "config3")

    # ---- multi-GPU self-validation (N > 1, outside every timed region): rank 0 recomputes the shard of the LAST rank on its own GPU (same
    # per-GPU batch, same kernels, inputs a function of the global sample ids) for a short clip and requires the all-gathered latents of that
    # shard to be EQUAL bit for bit — wrong gather order, wrong sharding or a rank-dependent result fails the run instead of passing silently ----
    shard_check = None
    if world > 1:
        from gtav_amd.generate import shard_inputs as _si
        ct, cs = n_prompt + 2, 3
        _, fr_me, nz_me = _si(world * B, rank, world, n_prompt, ct, (FH, FW), (LH, LW), seed=4242)
        xg, _ = generate_clip(dit, vae, fr_me.to(dev), nz_me.to(dev), ct, cs, None, ctx_cache=False) if vae is not None else (None, None)
        if xg is not None and rank == 0:
            other = world - 1
            _, fr_o, nz_o = _si(world * B, other, world, n_prompt, ct, (FH, FW), (LH, LW), seed=4242)
            xo, _ = generate_clip(dit, vae, fr_o.to(dev), nz_o.to(dev), ct, cs, None, ctx_cache=False, gather=False)
            same = bool(torch.equal(xg[other * B:(other + 1) * B], xo))
            shard_check = {"passed": same, "what": "latents of rank %d's shard (global samples %d..%d) as all-gathered == recomputed on rank 0, bit for bit; "
                                                   "%d-frame clip, %d noise steps" % (other, other * B, (other + 1) * B - 1, ct, cs),
                           "per_sample_abs_sum": [round(float(v), 6) for v in xg.abs().sum(dim=(1, 2, 3, 4)).cpu()]}
            if not same:
                sys.stderr.write("bench.py: SHARD SELF-CHECK FAILED: the gathered latents of rank %d differ from rank 0's recomputation\n" % other)
        dist.barrier()       # (the other ranks only took part in the gather)
        leg_done("shard_self_check")
    # ---- N > 1: what the path's one collective costs (outside the timed regions): all-gather of a clip's latents, and of the batched leg's ----
    multi_gpu = None
    if world > 1:
        from gtav_amd.generate import all_gather_latents
        ag = {}
        for name, b in (("headline", B), ("batched", Bb)):
            if b <= 0 or name in ag:
                continue
            xl = torch.zeros(b, total, 16, LH, LW, device=dev)
            all_gather_latents(xl)
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(5):
                all_gather_latents(xl)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            ag[name] = {"bytes_per_rank": xl.numel() * 4, "bytes_gathered": xl.numel() * 4 * world, "ms": round(ms, 3),
                        "bus_gbps": round((world - 1) / world * xl.numel() * 4 * world / (ms * 1e-3) / 1e9, 1)}
            del xl
        multi_gpu = {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(), "ranks_per_gpu": "one process per GPU (LOCAL_RANK -> cuda:LOCAL_RANK)",
                     "all_gather_latents": ag, "collectives_per_clip": 1}
        leg_done("collective_probe")

    # ---- bounded config4 leg (every N): BASELINE configs[4] AS SPECIFIED — batch 16 per GPU of 5-frame 360x640 clips, `encode_frames` of the 80 frames
    # INSIDE the timed region (train_dit.py:329-351,570), then forward + loss (train_dit.py:590-650) and, as a second timing, the whole optimisation step
    # (backward, bucketed gradient all-reduce over RCCL when N > 1 — overlapped with the backward pass —, clip, AdamW: SURVEY.md 8(f)1).  Frames are
    # resident in HBM before the region starts; barrier + synchronize on both sides, maximum over ranks ----
    config4 = None
    if args.config4_steps > 0 and args.geometry == "native" and fits("config4", 30.0 + 0.4 * args.config4_steps):
        config4 = bench_config4(args, world, rank, dev, dist, torch)
        leg_done("config4")

    # ---- bounded g256 leg (N = 1, native default run only): BASELINE.json's literal "[B, 32 frames, 256x256]" through the SURVEY.md 8(d) preset
    # (VAE patch 16 -> 16x16x16 latents, 64 DiT tokens per frame, DiT-S / ViT-L widths), batch 1, no actions, window algorithm, VAE in the timed region ----
    g256_leg = None
    if world == 1 and args.geometry == "native" and args.g256_clips > 0 and vae is not None and fits("g256", 12.0 + 5.0 * args.g256_clips):
        g2 = GEOM["g256"]
        dit2 = DiT(**g2["dit"], init_weights=False, max_batch=1)
        dit2.load_state_dict(W.synth_state_dict(W.dit_param_shapes(**g2["dit"]), seed=0))
        dit2.reserve(1, 5, steps)
        vae2 = AutoencoderKL(**g2["vae"], init_weights=False, max_frames_per_call=32)
        vae2.load_state_dict(W.synth_state_dict(W.vae_param_shapes(**g2["vae"]), seed=1))
        _, fr2, nz2 = shard_inputs(1, 0, 1, n_prompt, total, g2["frame"], g2["lat"], seed=1000)
        fr2, nz2 = fr2.to(dev), nz2.to(dev)
        pf2 = None
        if 256 <= 5 * 64 <= 1536:
            from gtav_amd.generate import tune_weight_prefetch
            pf2 = tune_weight_prefetch(dit2, 1, latent_hw=g2["lat"])
        generate_clip(dit2, vae2, fr2, nz2, total, 2, None, ctx_cache=False)          # short warm-up: same shapes / graph keys
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.g256_clips):
            generate_clip(dit2, vae2, fr2, nz2, total, steps, None, ctx_cache=False)
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t0
        p_save, P_TOK = P_TOK, 64
        fl2 = clip_flops(1, n_prompt, total, steps, 5, False)
        P_TOK = p_save
        g256_leg = {"workload": "BASELINE.json's literal 256x256 frames: SURVEY.md 8(d) g256 preset (VAE patch 16 -> 16x16x16 latents, 64 DiT tokens per frame, "
                                "DiT-S/2 widths depth 16 + ViT-L/16 VAE), %d frames (%d prompt), %d noise steps, batch 1, no actions, window algorithm, VAE in the "
                                "timed region" % (total, n_prompt, steps),
                    "clips_timed": args.g256_clips, "value": round((total - n_prompt) * args.g256_clips / e2, 4), "unit": "generated frames/s",
                    "ms_per_clip": round(e2 / args.g256_clips * 1e3, 1), "executed_pflop_per_clip": round(fl2 / 1e15, 4),
                    "achieved_tflops_per_gpu": round(fl2 * args.g256_clips / e2 / 1e12, 1), "weight_prefetch": pf2}
        del dit2, vae2, fr2, nz2
        torch.cuda.empty_cache()
        leg_done("g256")

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
                img = inp[0][0, :2].cpu() * 2 - 1
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

    leg_done("cpu_baseline")
    if rank == 0:
        cfg_name = ("DiT-S/2 (608M) + ViT-L/20 VAE, native 360x640 frames -> 16x18x32 latents" if args.geometry == "native" else
                    "DiT-S/2 widths on a 16x16 latent grid + ViT-L/16 VAE, g256 preset: 256x256 frames -> 16x16x16 latents (SURVEY.md 8(d))")
        line = {
            "metric": "generated frames/sec (32-frame clip, 100 noise steps)", "value": round(value, 4), "unit": "generated frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp16 (fp32 accumulate/residual)",
            "data": "synthetic (seeded CPU-generated prompt frames + noise by global sample id, hash-seeded random weights incl. adaLN)" +
                    (" — REHEARSAL: all ranks on one GPU over gloo, the numbers mean nothing" if os.environ.get("GTAV_BENCH_REHEARSE_ONE_GPU") == "1" and world > 1 else ""),
            "config": {"workload": "BASELINE configs[%d]: %s, %d frames (%d prompt), %d noise steps, batch %d per GPU%s" %
                                   (2 if args.use_actions else 1, cfg_name, total, n_prompt, steps, B,
                                    ", action-conditioned" if args.use_actions else ", no actions"),
                       "global_batch": world * B, "algorithm": "window-recompute (reference behaviour)" if head == "window" else "ctx-cached",
                       "parallelism": "batch-sharded x%d (one process per GPU), one all-gather of latents per clip" % world,
                       "vae_in_timed_region": vae is not None, "geometry": args.geometry},
            "all_frames_per_s": round(world * B * total * args.steps / el, 4),          # B * 32 / wall
            "dit_forwards_per_s": round(world * (total - n_prompt) * (steps + 1) * args.steps / el, 2),   # batched forwards of B samples
            "roofline": roofline, "cpu_baseline": cpu, "dit_step": dit_step,
            "roofline_cached": roofline_cached, "dit_step_cached": dit_step_cached, "weight_prefetch": prefetch_choice,
        }
        line.update(algo_report(B, results))
        line["timer"] = timer_calibration()
        line["wall"] = {"total_s": round(time.time() - T_START, 1), "budget_s": args.wall_budget, "legs_s": leg_wall, "skipped_legs": skipped_legs}
        if batched is not None:
            line["config2" if world == 1 else "config3"] = batched
        if config4 is not None:
            line["config4"] = config4
        if g256_leg is not None:
            line["g256"] = g256_leg
        if shard_check is not None:
            line["shard_self_check"] = shard_check
        if multi_gpu is not None:
            line["multi_gpu"] = multi_gpu
        print(json.dumps(line))


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank_main(args)


if __name__ == "__main__":
    main()
