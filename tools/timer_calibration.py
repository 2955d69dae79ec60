#!/usr/bin/env python3
"""Calibration of bench.py's in-situ timer against rocprofv3 (GPU box; driven by tools/timer_calibration.sh).

  run   : gtav_timer_calibrate at 5 us and 20 us spins (64 back-to-back launches of a one-wave kernel that spins on the device's 100 MHz clock, an event pair
          attached to each dispatch like a profiled kernel) -> JSON on stdout: what the event pairs read and what the kernel measured itself
  parse : the same process's rocprofv3 kernel trace (one row per dispatch) + that JSON -> profiles/timer_calibration.json:
          rocprof_minus_device_us = rocprofv3's duration of the spin kernel less the kernel's own figure = the dispatch ramp BOTH scales contain;
          bench.py subtracts (event_minus_device_us - rocprof_minus_device_us) from every per-launch reading, which puts its roofline on rocprofv3's scale.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import ctypes as C
    import torch
    from gtav_amd import lib as L
    torch.cuda.init()
    out = {}
    for spin in (5, 20):
        e, d = C.c_double(0), C.c_double(0)
        L.check(L.load().gtav_timer_calibrate(spin, 64, C.byref(e), C.byref(d), torch.cuda.current_stream().cuda_stream))
        out[str(spin)] = {"event_us": e.value, "device_us": d.value}
    print(json.dumps(out))


def parse(trace_dir, run_json, dst):
    runs = json.load(open(run_json))
    rows = []
    for f in glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "calib_spin_kernel" in r.get("Kernel_Name", ""):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort()
    assert len(rows) == 2 * 65, f"expected 130 spin launches (2 x (1 warm-up + 64)), found {len(rows)}"
    res = {}
    for i, spin in enumerate(("5", "20")):
        grp = rows[i * 65 + 1:(i + 1) * 65]                     # without the warm-up launch
        dur = sum(e - s for s, e in grp) / len(grp) / 1e3       # ns -> us
        res[spin] = {"rocprof_us": round(dur, 3), "device_us": round(runs[spin]["device_us"], 3), "event_us_under_rocprof": round(runs[spin]["event_us"], 3)}
    ramp = sum(v["rocprof_us"] - v["device_us"] for v in res.values()) / len(res)
    out = {"rocprof_minus_device_us": round(ramp, 3), "per_spin": res,
           "source": "tools/timer_calibration.sh: rocprofv3 --kernel-trace of tools/timer_calibration.py run (calib_spin_kernel, 2 x 64 launches); round 6"}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        parse(sys.argv[2], sys.argv[3], sys.argv[4])
