#!/bin/bash
# rocprofv3 evidence of a round (GPU box, from the repo root): kernel-trace statistics of (A) the batch-8 action-conditioned window run (BASELINE configs[2]),
# (B) the same batch with the exact context-cached algorithm (M = 1152 steps), (C) the batch-1 window run (BASELINE configs[1]); then the three PMC passes on the
# torch-free GEMM driver (tools/gemm_traffic.sh -> profiles/traffic.json).  Everything lands under gpurun_out/prof/ (copy the summaries to profiles/roundN/).
#   usage: bash tools/profile_round.sh
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
COMMON="--no-cpu-baseline --batched-clips 0 --train-leg-steps 0 --g256-clips 0 --warmup 0"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/config2_window" -o run -- python3 "$ROOT/bench.py" --batch-per-gpu 8 --use-actions --algo window --steps 1 $COMMON > "$OUT/config2_window.json" 2> "$OUT/config2_window.err"
echo "config2 window done" 
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/config2_cached" -o run -- python3 "$ROOT/bench.py" --batch-per-gpu 8 --use-actions --algo cached --steps 1 $COMMON > "$OUT/config2_cached.json" 2> "$OUT/config2_cached.err"
echo "config2 cached done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/config1_window" -o run -- python3 "$ROOT/bench.py" --algo window --steps 2 $COMMON > "$OUT/config1_window.json" 2> "$OUT/config1_window.err"
echo "config1 window done"
cd "$ROOT"
# keep only the statistics (the raw traces are hundreds of MB)
find "$OUT" -name "*kernel_trace.csv" -delete
bash tools/gemm_traffic.sh "$OUT/pmc"
ls -R "$OUT" | head -50
