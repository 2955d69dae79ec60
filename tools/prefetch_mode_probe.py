"""Captured batch-1 window step under a dozen settings of the next-weight L2 prefetch (gtav_dit_set_weight_prefetch's per-class modes: each of the four
weights — out-proj, fc1, fc2, to_qkv — skipped (0), whole slice (1) or its first 4 K tiles (4)), alternated in one process, then what
generate.tune_weight_prefetch picks.  Two lines of JSON.  (Round 5 first ran it with the prefetching blocks' slice shifted by 1..7 XCDs: the gain did not
change where the prefetch pays, so it is not the XCD-local L2 — profiles/round5/prefetch_box_survey.txt.)
Usage (GPU box): python tools/prefetch_mode_probe.py [--rounds 3]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--window", type=int, default=0, help="frames in the window (0 = the model's maximum); 1 = the shape of a context-cached step")
    ap.add_argument("--exp", action="store_true", help="experiments build (reads GTAV_PF_MIN_M: the smallest token count that prefetches)")
    a = ap.parse_args()
    if a.exp:
        from gtav_amd import lib as L
        L.load_experiments()
    import gtav_amd.weights as W
    from gtav_amd.model.dit import DiT
    from gtav_amd.generate import _alphas_cumprod
    dev = torch.device("cuda", 0)
    model = DiT(depth=16, init_weights=False, max_batch=a.batch)
    model.load_state_dict(W.synth_state_dict(W.dit_param_shapes(depth=16), seed=0))
    B, T = a.batch, (a.window or model.max_frames)
    x0 = (torch.randn(B, T, model.in_channels, model.input_h, model.input_w, generator=torch.Generator().manual_seed(11)) * 0.5).to(dev)
    model.set_schedule(_alphas_cumprod(1e-4))
    steps = a.steps
    ts = [999 - 7 * k for k in range(steps + 3)]
    from gtav_amd.generate import prefetch_mode as PM
    # (out, fc1, fc2, qkv): off, on, first 4 K tiles of each, then single classes skipped, then pairs
    combos = [(0, 0, 0, 0), (1, 1, 1, 1), (4, 4, 4, 4), (0, 1, 1, 1), (1, 0, 1, 1), (1, 1, 0, 1), (1, 1, 1, 0), (1, 0, 0, 1), (1, 4, 4, 1), (1, 0, 4, 1), (0, 0, 0, 1), (1, 0, 0, 0)]
    modes = [PM(c) for c in combos]
    best = {m: float("inf") for m in modes}
    with torch.inference_mode():
        for _ in range(a.rounds):
            for mode in modes:
                model.set_weight_prefetch(mode)
                x = x0.clone()
                model.prepare_frame_(B, T, 0, T - 1, 15, ts, None)
                for k in range(3):
                    model.denoise_step_(x, 0, T - 1, 15, ts[k], ts[k + 1], False, None, cond_step=k)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for k in range(3, steps + 3):
                    model.denoise_step_(x, 0, T - 1, 15, ts[k], ts[min(k + 1, steps + 2)], False, None, cond_step=k)
                torch.cuda.synchronize(dev)
                best[mode] = min(best[mode], (time.perf_counter() - t0) / steps * 1e3)
    model.check()
    from gtav_amd.generate import tune_weight_prefetch
    print(json.dumps({"tuner": tune_weight_prefetch(model, B, window=T)}))
    print(json.dumps({"uuid": torch.cuda.get_device_properties(0).uuid.__str__() if hasattr(torch.cuda.get_device_properties(0), "uuid") else "",
                      "ms_by_out_fc1_fc2_qkv": {"".join(str(v) for v in c): round(best[PM(c)], 4) for c in combos}}))


if __name__ == "__main__":
    main()
