#!/bin/bash
# rocprofv3 counter passes on the long-sequence attention launch (tools/attn_bench.py, S = 576, 80 frames x 16 heads): where a wave's cycles go.
#   usage (GPU box, repo root): bash tools/attn_pmc.sh [outdir]
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=${1:-$ROOT/gpurun_out/attn_pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE"
P3="SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS"
rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d "$OUT/p1" -- python3 "$ROOT/tools/attn_bench.py" --s 576 --nb 80 --iters 10 > "$OUT/p1.log" 2>&1
rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d "$OUT/p2" -- python3 "$ROOT/tools/attn_bench.py" --s 576 --nb 80 --iters 10 > "$OUT/p2.log" 2>&1
rocprofv3 --pmc $P3 --kernel-trace --output-format csv -d "$OUT/p3" -- python3 "$ROOT/tools/attn_bench.py" --s 576 --nb 80 --iters 10 > "$OUT/p3.log" 2>&1 || echo "pass 3 failed (a counter name may not exist on this build)"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("p1", "p2", "p3"):
    files = glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if "attn_flash" in r["Kernel_Name"] or "attn_spatial" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        v = v[len(v) // 2:]      # drop warm-up launches
        print(f"{p} {k:32s} mean per launch {sum(v) / len(v):16.1f}  ({len(v)} launches)")
PY
