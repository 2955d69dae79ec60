"""VAE encode at the TRAINER's batch (train_dit.py:329-351: 16 clips x 5 frames = 80 frames of 360x640 per step; BASELINE configs[4]): wall time per
encode of 80 frames for several frames-per-call settings, product library.  Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel table
(profiles/round5/).  Usage (GPU box): python tools/vae_encode_profile.py [--frames 80] [--per-call 40,80] [--reps 5]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402

ENC_GFLOP_PER_FRAME = 96.6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=80)
    ap.add_argument("--per-call", default="40,80")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    L.load()
    import gtav_amd.weights as W
    from gtav_amd.model.vae import VAE_models
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    img = (torch.rand(args.frames, 3, 360, 640, generator=g) * 2 - 1).to(dev)
    sd = W.synth_state_dict(W.vae_param_shapes(), seed=1)
    for pc in [int(v) for v in args.per_call.split(",")]:
        vae = VAE_models["vit-l-20-shallow-encoder"](init_weights=False, max_frames_per_call=pc)
        vae.load_state_dict(sd)
        for _ in range(2):
            vae.encode_moments(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            vae.encode_moments(img)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.reps * 1e3
        tf = args.frames * ENC_GFLOP_PER_FRAME / ms
        print(f"encode {args.frames} frames, {pc} per call: {ms:.2f} ms = {tf:.0f} TFLOP/s = {tf / 2500:.3f} of the MFMA peak", flush=True)
        vae.check()
        del vae
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
