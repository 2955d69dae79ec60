#!/bin/bash
# Round-6 rocprofv3 evidence (GPU box, from the repo root), VERDICT r5 item 1: for each of BASELINE configs[1] (batch 1, window), configs[2] window and configs[2]
# context-cached — (a) the leg WITHOUT the profiler (its in-situ per-class times), (b) the same command under `rocprofv3 --kernel-trace --stats`; the trace is reduced
# to per-class averages (tools/kernel_trace_classes.py) beside the in-situ figures of (a), same box, same build.  Output: gpurun_out/prof6/ (copy to profiles/round6/).
#   usage: bash tools/profile_round6.sh [legs: c1 c2w c2c]
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof6
mkdir -p "$OUT"
export TMPDIR=/tmp
LEGS=${@:-c1 c2w c2c}
COMMON="--no-cpu-baseline --config4-steps 0 --g256-clips 0 --warmup 0"
for leg in $LEGS; do
  case $leg in
    c1)  ARGS="--algo both --steps 2 --cached-clips 1 --batched-clips 0"; LEGNAME=headline ;;
    c2w) ARGS="--batch-per-gpu 8 --use-actions --algo window --steps 1 --batched-clips 0"; LEGNAME=headline ;;
    c2c) ARGS="--batch-per-gpu 8 --use-actions --algo cached --steps 2 --batched-clips 0"; LEGNAME=headline ;;
  esac
  cd /tmp
  python3 "$ROOT/bench.py" $ARGS $COMMON > "$OUT/${leg}_plain.json" 2> "$OUT/${leg}_plain.err"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$leg" -o run -- python3 "$ROOT/bench.py" $ARGS $COMMON > "$OUT/${leg}_under_rocprof.json" 2> "$OUT/${leg}_under_rocprof.err"
  cd "$ROOT"
  python3 tools/kernel_trace_classes.py "$OUT/$leg" --insitu "$OUT/${leg}_plain.json" --leg $LEGNAME --out "$OUT/${leg}_per_class.json" > /dev/null
  find "$OUT/$leg" -name "*kernel_trace.csv" -delete
  echo "$leg done"
done
ls -R "$OUT" | head -40
