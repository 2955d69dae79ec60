"""Second half of tools/gemm_traffic.sh: rocprofv3 counter_collection csv files -> profiles/traffic.json."""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, K = 4096, 1024


def per_launch(outdir, counter):
    """mean counter value (KB) per GEMM dispatch, keyed by grid size (the two M values launch different grids)"""
    files = glob.glob(os.path.join(outdir, counter, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {outdir}/{counter}"
    acc = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if "gemm" not in row["Kernel_Name"] or row["Counter_Name"] != counter:
                continue
            key = (int(row["Grid_Size"]), row["Kernel_Name"])
            a = acc.setdefault(key, [0.0, 0])
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


def main():
    outdir = sys.argv[1]
    fetch, write = per_launch(outdir, "FETCH_SIZE"), per_launch(outdir, "WRITE_SIZE")
    sha = hashlib.sha256(open(os.path.join(ROOT, "ai-generated-gtav_amd", "csrc", "gemm.hip"), "rb").read()).hexdigest()[:16]
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only) on tools/gemm_pmc via tools/gemm_traffic.sh; "
                     "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 tallies 128-B read requests as 64 B, MI355X_MICROARCH.md HBM section)",
           "gemm_hip_sha16": sha}
    keys = sorted(fetch.keys())          # smaller grid = M 720, larger = M 5760
    assert len(keys) == 2, f"expected the two fc1 launches, found {keys}"
    for (key, M) in zip(keys, (720, 5760)):
        f_kb, nf = fetch[key]
        w_kb, nw = write[key]
        res["fc1_M%d" % M] = {"hbm_bytes_per_launch": int((2 * f_kb + w_kb) * 1024), "fetch_size_kb": round(f_kb, 1), "write_size_kb": round(w_kb, 1),
                               "launches": min(nf, nw), "kernel": key[1], "grid_threads": key[0],
                               "algorithmic_bytes": N * K * 2 + M * K * 2 + M * N * 2}
    json.dump(res, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
