#!/bin/bash
# Builds libgtav_amd.so (gfx950) in-tree. Usage: csrc/build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
OUT=../libgtav_amd.so
mkdir -p /tmp/gtav_build
pids=()
for f in gemm skinny elementwise attention api; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed "$@" -c $f.hip -o /tmp/gtav_build/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT /tmp/gtav_build/{gemm,skinny,elementwise,attention,api}.o
echo "built $(realpath $OUT)"
