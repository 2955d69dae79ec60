"""A/B: ONE handle stepping a batch of B clips against TWO handles of B / 2 clips each on two HIP streams (same weights, same inputs).
The kernels of a launch are bulk-synchronous — every block is in its prologue, its MFMA loop or its store burst at the same time — so memory
phases and matrix phases of a step add up (docs/LABNOTES.md 4.7).  Two half-batches on two streams put different kernels on the chip at the same time.
Product library.  Usage (GPU box): python tools/two_stream_ab.py [--batch 8] [--cached] [--steps 30] [--rounds 3]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from gtav_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--cached", action="store_true")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    L.load()
    import gtav_amd.weights as W
    from gtav_amd.generate import _alphas_cumprod
    from gtav_amd.model.dit import DiT_models
    dev = torch.device("cuda", 0)
    B, P = a.batch, a.parts
    assert B % P == 0
    sd = W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0)
    F = 6
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(B, F, 16, 18, 32, generator=g) * 0.5).to(dev)
    nr = torch.linspace(0, 999, a.steps + 1)
    t_of = [int(v) for v in nr]
    order = list(reversed(range(a.steps + 1)))
    i, start = 5, 1

    def make(b):
        d = DiT_models["DiT-S/2"](init_weights=False, max_batch=b)
        d.load_state_dict(sd)
        d.reserve(b, 5, a.steps)
        d.set_schedule(_alphas_cumprod(1e-4))
        return d

    whole = make(B)
    parts = [make(B // P) for _ in range(P)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
    xw = torch.empty_like(x)
    xp = [torch.empty_like(x[k * (B // P):(k + 1) * (B // P)]) for k in range(P)]

    def run_whole():
        xw.copy_(x)
        whole.prepare_frame_(B, F, start, i, 15, [t_of[k] for k in order], None)
        t0 = None
        for step, ni in enumerate(order):
            if step == 1:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            whole.denoise_step_(xw, start, i, 15, t_of[ni], t_of[max(0, ni - 1)], ni <= 0, None, cached=a.cached and step > 0, cond_step=step)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (len(order) - 1) * 1e3

    def run_parts():
        b = B // P
        for k in range(P):
            xp[k].copy_(x[k * b:(k + 1) * b])
            parts[k].prepare_frame_(b, F, start, i, 15, [t_of[q] for q in order], None)
        torch.cuda.synchronize()
        t0 = None
        for step, ni in enumerate(order):
            if step == 1:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            for k in range(P):
                with torch.cuda.stream(streams[k]):
                    parts[k].denoise_step_(xp[k], start, i, 15, t_of[ni], t_of[max(0, ni - 1)], ni <= 0, None, cached=a.cached and step > 0, cond_step=step)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (len(order) - 1) * 1e3

    out = {"batch": B, "parts": P, "algo": "cached" if a.cached else "window", "whole_ms_per_step": [], "parts_ms_per_step": []}
    run_whole(); run_parts()
    for _ in range(a.rounds):
        out["whole_ms_per_step"].append(round(run_whole(), 4))
        out["parts_ms_per_step"].append(round(run_parts(), 4))
    diff = (torch.cat(xp).float() - xw.float()).norm() / xw.float().norm()
    out["rel_l2_parts_vs_whole"] = float(diff)
    print(json.dumps(out))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
