#!/bin/bash
# Calibrates bench.py's in-situ timer against rocprofv3 (GPU box, from the repo root): runs tools/timer_calibration.py under `rocprofv3 --kernel-trace` and once
# without the profiler, writes profiles/timer_calibration.json (copied to gpurun_out/ for the trip home).  See the .py for what is measured.
#   usage: bash tools/timer_calibration.sh
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/timer
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
python3 "$ROOT/tools/timer_calibration.py" run > "$OUT/plain_run.json"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o run -- python3 "$ROOT/tools/timer_calibration.py" run > "$OUT/rocprof_run.json" 2> "$OUT/rocprof_run.err"
cd "$ROOT"
python3 tools/timer_calibration.py parse "$OUT/trace" "$OUT/rocprof_run.json" "$OUT/timer_calibration.json"
cp "$OUT/timer_calibration.json" profiles/timer_calibration.json
echo "plain run (no profiler): $(cat "$OUT/plain_run.json")"
find "$OUT/trace" -name "*kernel_trace.csv" -delete
