#!/bin/bash
# Regenerates profiles/traffic.json (bench.py's `roofline.traffic` / `roofline.mfma_busy`) from live PMC passes on the GPU box:
#   one `rocprofv3 --pmc FETCH_SIZE` pass and one `--pmc WRITE_SIZE` pass (they do not fit one pass: TCC has 4 slots, FETCH_SIZE
#   takes 3, WRITE_SIZE 2 — MI355X_MICROARCH.md "rocprofv3 PMC slots") and one `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE`
#   pass, kernel trace only, on the torch-free driver tools/gemm_pmc (the four GEMM classes of a DiT half-block through the C-ABI at M = 720
#   and M = 5760, rotating weight buffers);
#   bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 — gfx950 tallies 128-B read requests as 64 B (same guide, "HBM").
# The json records the sha of csrc/gemm.hip it was measured on; bench.py reports `traffic: null` for any other build.
#   usage (GPU box, from the repo root):  bash tools/gemm_traffic.sh [outdir]      (outdir defaults to gpurun_out/pmc)
set -e
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
hipcc -O2 tools/gemm_pmc.cpp -Iinclude -L ai-generated-gtav_amd -lgtav_amd -Wl,-rpath,"$PWD/ai-generated-gtav_amd" -o tools/gemm_pmc
rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE" "$OUT/MFMA"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/FETCH_SIZE" -- ./tools/gemm_pmc 16 > "$OUT/FETCH_SIZE.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/WRITE_SIZE" -- ./tools/gemm_pmc 16 > "$OUT/WRITE_SIZE.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/MFMA" -- ./tools/gemm_pmc 16 > "$OUT/MFMA.log" 2>&1
python3 tools/gemm_traffic.py "$OUT"
cp profiles/traffic.json gpurun_out/traffic.json   # the GPU box only sends gpurun_out/ back: copy it to profiles/ in the repo
