#!/bin/bash
# Final round-5 rocprofv3 evidence on the end-of-round build (GPU box, repo root): kernel statistics of the VAE encode of 80 frames, of BASELINE configs[4]
# forward + loss and of the whole optimisation step, then a default bench.py line.  Output: gpurun_out/prof5f/
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof5f
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
python3 "$ROOT/tools/vae_encode_profile.py" --per-call 80 --reps 5 > "$OUT/vae_encode.txt" 2> "$OUT/vae_encode.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/vae" -o run -- python3 "$ROOT/tools/vae_encode_profile.py" --per-call 80 --reps 5 > "$OUT/vae_encode_under_rocprof.txt" 2>> "$OUT/vae_encode.err"
echo "vae done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/config4_forward" -o run -- python3 "$ROOT/bench.py" --mode train --steps 3 --warmup 1 > "$OUT/config4_forward.json" 2> "$OUT/config4_forward.err"
echo "config4 forward done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/config4_step" -o run -- python3 "$ROOT/bench.py" --mode train_step --steps 3 --warmup 1 > "$OUT/config4_step.json" 2> "$OUT/config4_step.err"
echo "config4 step done"
cd "$ROOT"
find "$OUT" -name "*kernel_trace.csv" -delete
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "bench done"
ls -R "$OUT" | head -40
