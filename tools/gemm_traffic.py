"""Second half of tools/gemm_traffic.sh: rocprofv3 counter_collection csv files -> profiles/traffic.json.
tools/gemm_pmc dispatches, for M in (720, 1152, 5760, 11520): qkv, out, fc1, fc2, qkvs — ITERS launches each; the rows are segmented by dispatch order."""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ITERS = 16
CLASSES = [("qkv", 3072, 1024), ("out", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096), ("qkvs", 3072, 1024)]   # qkvs: the fused spatial to_qkv + attention launch (gemm_qkvs_attn_kernel): reads X and W, writes the attention output [M][1024] fp16
MS = (720, 1152, 5760, 11520)
VAE_M = 46080
VAE_CLASSES = [("vae_qkv", 3072, 1024), ("vae_proj", 1024, 1024), ("vae_fc1", 4096, 1024), ("vae_fc2", 1024, 4096)]


def rows_by_dispatch(outdir, sub, counters):
    """{counter: [(dispatch id, kernel, grid, value)] sorted by dispatch id} of the GEMM dispatches"""
    files = glob.glob(os.path.join(outdir, sub, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {outdir}/{sub}"
    out = {c: {} for c in counters}
    for f in files:
        for row in csv.DictReader(open(f)):
            if "gemm" not in row["Kernel_Name"] or row["Counter_Name"] not in out:
                continue
            d = out[row["Counter_Name"]].setdefault(int(row["Dispatch_Id"]), [row["Kernel_Name"], int(row["Grid_Size"]), 0.0])
            d[2] += float(row["Counter_Value"])          # (a counter may be reported per XCD / SE: sum the rows of one dispatch)
    return {c: [(k,) + tuple(v) for k, v in sorted(d.items())] for c, d in out.items()}


def segments(rows):
    nexp = ITERS * (len(CLASSES) * len(MS) + len(VAE_CLASSES))
    assert len(rows) == nexp, f"expected {nexp} GEMM dispatches, found {len(rows)}"
    seg = {}
    i = 0
    for M in MS:
        for name, N, K in CLASSES:
            chunk = rows[i: i + ITERS]
            i += ITERS
            assert len({r[1] for r in chunk}) == 1, f"{name} M={M}: mixed kernels in one segment"
            seg[(name, M)] = (sum(r[3] for r in chunk) / ITERS, chunk[0][1], chunk[0][2])
    for name, N, K in VAE_CLASSES:
        chunk = rows[i: i + ITERS]
        i += ITERS
        assert len({r[1] for r in chunk}) == 1, f"{name}: mixed kernels in one segment"
        seg[(name, VAE_M)] = (sum(r[3] for r in chunk) / ITERS, chunk[0][1], chunk[0][2])
    return seg


def main():
    outdir = sys.argv[1]
    fetch = segments(rows_by_dispatch(outdir, "FETCH_SIZE", ["FETCH_SIZE"])["FETCH_SIZE"])
    write = segments(rows_by_dispatch(outdir, "WRITE_SIZE", ["WRITE_SIZE"])["WRITE_SIZE"])
    mf = rows_by_dispatch(outdir, "MFMA", ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])
    mfma, sqb, gui = (segments(mf[c]) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))
    sha = hashlib.sha256(open(os.path.join(ROOT, "ai-generated-gtav_amd", "csrc", "gemm.hip"), "rb").read()).hexdigest()[:16]
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (separate passes, kernel trace "
                     "only) on tools/gemm_pmc via tools/gemm_traffic.sh; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 tallies 128-B read requests as 64 B, "
                     "MI355X_MICROARCH.md HBM section); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): the fraction of the "
                     "launch's shader-clock cycles in which a SIMD's matrix pipe was busy",
           "gemm_hip_sha16": sha}
    for M, name, N, K in [(M_, n_, N_, K_) for M_ in MS for (n_, N_, K_) in CLASSES] + [(VAE_M, n_, N_, K_) for (n_, N_, K_) in VAE_CLASSES]:
        if True:
            f_kb, kern, grid = fetch[(name, M)]
            w_kb = write[(name, M)][0]
            cyc = gui[(name, M)][0] / 8.0
            splitk = 1
            out_bytes = M * N * (2 if name in ("qkv", "fc1", "vae_qkv", "vae_fc1") else 4)
            if name == "qkvs":
                out_bytes = M * 1024 * 2
            res[f"{name}_M{M}"] = {"hbm_bytes_per_launch": int((2 * f_kb + w_kb) * 1024), "fetch_size_kb": round(f_kb, 1), "write_size_kb": round(w_kb, 1),
                                   "launches": ITERS, "kernel": kern, "grid_threads": grid,
                                   # (the in-place residual epilogue of the large-M out-proj / fc2 — EPI_RESID, gemm_lp_kernel<4, ..> — also READS its fp32 output tile)
                                   "algorithmic_bytes": N * K * 2 + M * K * 2 + out_bytes * (2 if (name in ("out", "fc2") and "gemm_lp_kernel<4" in kern) or name in ("vae_proj", "vae_fc2") else 1),
                                   "mfma_busy_cycles": round(mfma[(name, M)][0]), "sq_busy_cycles": round(sqb[(name, M)][0]),
                                   "shader_cycles": round(cyc), "mfma_busy": round(mfma[(name, M)][0] / (1024.0 * cyc), 4) if cyc > 0 else None}
    json.dump(res, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
