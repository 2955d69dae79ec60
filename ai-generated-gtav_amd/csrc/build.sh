#!/bin/bash
# Builds libgtav_amd.so (gfx950) in-tree. Usage: csrc/build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
OUT=../libgtav_amd.so
mkdir -p /tmp/gtav_build
pids=()
for f in gemm skinny elementwise attention api; do
  extra=""
  # attention: keep MFMA accumulators in VGPRs — the softmax works on them every key block, and the AGPR form costs 56
  # v_accvgpr moves per 16 MFMAs there (the GEMMs touch their accumulators only in the epilogue and keep the default)
  [ $f = attention ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed $extra "$@" -c $f.hip -o /tmp/gtav_build/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT /tmp/gtav_build/{gemm,skinny,elementwise,attention,api}.o
echo "built $(realpath $OUT)"
