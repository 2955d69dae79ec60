#!/bin/bash
# Regenerates profiles/traffic.json (bench.py's `roofline.traffic`) from live PMC passes on the GPU box:
#   one `rocprofv3 --pmc FETCH_SIZE` pass and one `--pmc WRITE_SIZE` pass (they do not fit one pass: TCC has 4 slots, FETCH_SIZE
#   takes 3, WRITE_SIZE 2 — MI355X_MICROARCH.md "rocprofv3 PMC slots"), kernel trace only, on the torch-free driver tools/gemm_pmc
#   (fc1-shaped GEMM through the C-ABI at M = 720 and M = 5760, rotating weight buffers);
#   bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 — gfx950 tallies 128-B read requests as 64 B (same guide, "HBM").
# The json records the sha of csrc/gemm.hip it was measured on; bench.py reports `traffic: null` for any other build.
#   usage (GPU box, from the repo root):  bash tools/gemm_traffic.sh [outdir]      (outdir defaults to gpurun_out/pmc)
set -e
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
hipcc -O2 tools/gemm_pmc.cpp -Iinclude -L ai-generated-gtav_amd -lgtav_amd -Wl,-rpath,"$PWD/ai-generated-gtav_amd" -o tools/gemm_pmc
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$OUT/$c"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -- ./tools/gemm_pmc 16 > "$OUT/$c.log" 2>&1
done
python3 tools/gemm_traffic.py "$OUT"
cp profiles/traffic.json gpurun_out/traffic.json   # the GPU box only sends gpurun_out/ back: copy it to profiles/ in the repo
