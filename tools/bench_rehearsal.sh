#!/bin/bash
# Rehearses bench.py's N > 1 control flow on a ONE-GPU box: the launcher line the driver uses, two ranks, both on cuda:0, collectives over gloo
# (GTAV_BENCH_REHEARSE_ONE_GPU=1; RCCL refuses two ranks on one device).  Exercises sharding by global sample id, the all-gather of latents, the
# shard self-check, max-over-ranks timing and the rank-0-only JSON line.  The numbers mean nothing.   usage: bash tools/bench_rehearsal.sh [outdir]
set -e
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/rehearsal}
mkdir -p "$OUT"
export GTAV_BENCH_REHEARSE_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 1 --warmup 1 \
    --batched-clips 1 > "$OUT/bench_gpus2_generate.json" 2> "$OUT/bench_gpus2_generate.err"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --steps 2 --warmup 1 \
    --mode train_step > "$OUT/bench_gpus2_train_step.json" 2> "$OUT/bench_gpus2_train_step.err"
tail -c 600 "$OUT/bench_gpus2_generate.json"; echo; tail -c 600 "$OUT/bench_gpus2_train_step.json"; echo
