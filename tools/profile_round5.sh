#!/bin/bash
# Round-5 rocprofv3 evidence (GPU box, repo root): kernel-trace statistics of BASELINE configs[4] (bench.py --mode train: VAE encode of 80 frames + DiT forward + loss;
# --mode train_step: the whole optimisation step), then the PMC passes behind profiles/traffic.json (tools/gemm_traffic.sh).  Output: gpurun_out/prof5/
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof5
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/config4_forward" -o run -- python3 "$ROOT/bench.py" --mode train --steps 3 --warmup 1 > "$OUT/config4_forward.json" 2> "$OUT/config4_forward.err"
echo "config4 forward done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/config4_step" -o run -- python3 "$ROOT/bench.py" --mode train_step --steps 3 --warmup 1 > "$OUT/config4_step.json" 2> "$OUT/config4_step.err"
echo "config4 step done"
cd "$ROOT"
find "$OUT" -name "*kernel_trace.csv" -delete
bash tools/gemm_traffic.sh "$OUT/pmc"
ls -R "$OUT" | head -40
