"""Per-kernel-class in-situ times of ONE sampler step (window or context-cached) at a given batch, plus the captured-graph time of the same step.
Product library.  The classes come from the handle's profiler (HIP events attached to each dispatch, gtav_dit_profile); with the profiler on the
step runs eagerly, so the graph time is measured separately, profiler off.
Usage (GPU box): python tools/step_profile.py [--batch 8] [--cached] [--actions] [--steps 30]"""
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
    ap.add_argument("--actions", action="store_true")
    ap.add_argument("--steps", type=int, default=30, help="noise steps per generated frame in the measurement")
    ap.add_argument("--exp", action="store_true", help="experiments build (GTAV_* environment overrides)")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    if a.exp:
        L.load_experiments()
    else:
        L.load()
    import gtav_amd.weights as W
    from gtav_amd.generate import _alphas_cumprod
    from gtav_amd.model.dit import DiT_models
    dev = torch.device("cuda", 0)
    B = a.batch
    dit = DiT_models["DiT-S/2"](init_weights=False, max_batch=B)
    dit.load_state_dict(W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0))
    dit.reserve(B, 5, a.steps)
    g = torch.Generator().manual_seed(3)
    F = 6
    x = (torch.randn(B, F, 16, 18, 32, generator=g) * 0.5).to(dev)
    act = None
    if a.actions:
        act = torch.zeros(B, F, 25, device=dev)
        act[:, :, 3] = 1
    dit.set_schedule(_alphas_cumprod(1e-4))
    nr = torch.linspace(0, 999, a.steps + 1)
    t_of = [int(v) for v in nr]
    order = list(reversed(range(a.steps + 1)))
    i, start = 5, 1

    xbuf = torch.empty_like(x)          # ONE buffer: the captured graph is keyed by it

    def frame(count_from=1):
        """one generated frame: the first step always runs the whole window; returns (seconds, steps) of the steps from `count_from` on"""
        xx = xbuf
        xx.copy_(x)
        dit.prepare_frame_(B, F, start, i, 15, [t_of[k] for k in order], act)
        t0 = None
        for step, ni in enumerate(order):
            if step == count_from:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            dit.denoise_step_(xx, start, i, 15, t_of[ni], t_of[max(0, ni - 1)], ni <= 0, act, cached=a.cached and step > 0, cond_step=step)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, len(order) - count_from

    frame()
    frame()
    dt, n = frame()
    graph_ms = dt / n * 1e3
    # profiler on (eager launches): classes of the steps behind the first one
    xx = x.clone()
    dit.prepare_frame_(B, F, start, i, 15, [t_of[k] for k in order], act)
    dit.denoise_step_(xx, start, i, 15, t_of[order[0]], t_of[order[1]], False, act, cached=False, cond_step=0)
    torch.cuda.synchronize()
    dit.profile(True)
    nprof = 0
    for step, ni in enumerate(order):
        if step == 0:
            continue
        dit.denoise_step_(xx, start, i, 15, t_of[ni], t_of[max(0, ni - 1)], ni <= 0, act, cached=a.cached, cond_step=step)
        nprof += 1
    torch.cuda.synchronize()
    prof = dit.profile_read()
    dit.profile(False)
    M = B * 144 * (1 if a.cached else 5)
    D, HM = 1024, 4096
    gflop = {"gemm_qkv": 2.0 * M * 3 * D * D, "gemm_out": 2.0 * M * D * D, "gemm_fc1": 2.0 * M * HM * D, "gemm_fc2": 2.0 * M * HM * D}
    ev_ms, ev_n = prof.pop("empty_event_pair")
    out = {"batch": B, "algo": "cached" if a.cached else "window", "tokens_per_step": M, "graph_ms_per_step": round(graph_ms, 4), "classes": {}}
    tot = 0.0
    for k, (ms, cnt) in prof.items():
        e = {"ms_per_step": round(ms / nprof, 4), "launches_per_step": cnt // nprof}
        if cnt:
            e["us_per_launch"] = round(ms / cnt * 1e3, 2)
        if k in gflop and cnt:
            e["tflops"] = round(gflop[k] / (ms / cnt * 1e-3) / 1e12, 1)
            e["frac_of_mfma_peak"] = round(e["tflops"] / 2500.0, 4)
        out["classes"][k] = e
        tot += ms / nprof
    out["kernel_sum_ms_per_step"] = round(tot, 4)
    out["empty_event_pair_us"] = round(ev_ms / max(ev_n, 1) * 1e3, 2)
    print(json.dumps(out, indent=1))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
