#!/usr/bin/env python3
"""rocprofv3 kernel trace of a bench.py leg -> per-CLASS average kernel durations, beside the same leg's in-situ figures (the line's
us_per_launch must follow from profiles/).

rocprofv3's statistics are per kernel SYMBOL, and one symbol serves several classes (the split-K slab kernel runs out-proj and fc2, the persistent kernel both
in-place residual GEMMs): this walks the trace in dispatch order instead.  The model's big GEMMs cycle to_qkv -> out-proj -> fc1 -> fc2 (epilogue template
argument 5 | 4 or 6 | 2 or 3 | 4 or 6), which names every dispatch; LayerNorm / attention kernels are classes by symbol.  Dispatches are grouped by (class,
grid size) so that the legs of one process (the window step's M, the cached step's M, the VAE's) stay apart.

  usage: kernel_trace_classes.py <dir with *kernel_trace.csv> [--insitu plain_run.json --leg headline|config2] [--out table.json]"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys


def classify(rows):
    """rows: (start, dur_ns, name, grid) sorted by start -> list of (class, grid, dur_ns)"""
    out = []
    cyc = ["gemm_qkv", "gemm_out", "gemm_fc1", "gemm_fc2"]
    want = {"gemm_qkv": {5}, "gemm_out": {4, 6}, "gemm_fc1": {2, 3}, "gemm_fc2": {4, 6}}
    pos = 0
    bad = 0
    for _, dur, name, grid in rows:
        if "gemm_qkvs_attn_kernel" in name or "gemm_qkvt_attn_kernel" in name:   # the fused to_qkv + attention launches (template argument = ring stages, not an epilogue)
            pos = 1
            out.append(("attn_%s (the fused to_qkv + attention launch: the in-situ profiler books it under this class too)" % ("spatial" if "qkvs" in name else "temporal"), grid, dur))
            continue
        m = re.search(r"gemm\w*_kernel<(\d+)", name)
        if m and "grouped" not in name and "tn_kernel" not in name:
            epi = int(m.group(1))
            if epi in (2, 3, 4, 5, 6):
                if epi == 5:
                    pos = 0                      # re-synchronise on every to_qkv
                cls = cyc[pos % 4]
                if epi not in want[cls]:
                    bad += 1
                    cls = "gemm_other"
                else:
                    pos += 1
                out.append((cls, grid, dur))
                continue
            out.append(("gemm_f32_embed_final", grid, dur))
            continue
        if "ln_row_block_kernel" in name or "ln_wave_row_kernel" in name:
            out.append(("ln_modulate", grid, dur))
        elif "attn_spatial" in name or "attn_flash" in name:
            out.append(("attn_spatial", grid, dur))
        elif "attn_temporal" in name:
            out.append(("attn_temporal", grid, dur))
        elif "skinny" in name:
            out.append(("skinny_fp32_conditioning", grid, dur))
    return out, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace_dir")
    ap.add_argument("--insitu", default=None, help="JSON line of the same command run WITHOUT the profiler (in-situ per-class times)")
    ap.add_argument("--leg", default="headline", choices=["headline", "config2"])
    ap.add_argument("--out", default=None)
    ap.add_argument("--min-calls", type=int, default=200)
    a = ap.parse_args()
    rows = []
    for f in glob.glob(os.path.join(a.trace_dir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))
            rows.append((s, e - s, r["Kernel_Name"], grid))
    rows.sort()
    cl, bad = classify(rows)
    agg = collections.defaultdict(lambda: [0, 0])
    for cls, grid, dur in cl:
        v = agg[(cls, grid)]
        v[0] += 1
        v[1] += dur
    table = [{"class": c, "blocks": g, "calls": n, "avg_us": round(t / n / 1e3, 3)} for (c, g), (n, t) in sorted(agg.items()) if n >= a.min_calls]
    res = {"dispatches": len(rows), "unclassified_gemm_dispatches": bad, "rocprofv3_per_class": table}
    if a.insitu:
        line = json.loads(open(a.insitu).read().strip().splitlines()[-1])
        src = line if a.leg == "headline" else line.get("config2") or line.get("config3")
        cmp_ = {}
        for key, stepkey in (("dit_step", "window"), ("dit_step_cached", "cached")):
            ds = src.get(key)
            if not ds:
                continue
            for cname, v in ds["kernel_classes"].items():
                if "us_per_launch" in v:
                    us = v["us_per_launch"]
                else:
                    per = v.get("ms_per_forward", v.get("ms_per_step"))
                    n = v.get("launches_per_forward", v.get("launches_per_step"))
                    us = per / n * 1e3 if n else None
                cmp_[f"{stepkey}:{cname}"] = us
        res["in_situ_us_per_launch_same_box_no_profiler"] = cmp_
        res["timer"] = line.get("timer")
    js = json.dumps(res, indent=1)
    if a.out:
        open(a.out, "w").write(js)
    print(js)


if __name__ == "__main__":
    main()
