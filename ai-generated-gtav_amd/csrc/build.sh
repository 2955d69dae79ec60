#!/bin/bash
# Builds the HIP library (gfx950) in-tree.
#   csrc/build.sh [extra hipcc flags]       -> ../libgtav_amd.so       the product: no env knobs, no result-changing switches
#   csrc/build.sh exp [extra hipcc flags]   -> ../libgtav_amd_exp.so   -DGTAV_EXPERIMENTS, for tools/ only
# Objects go to a private mktemp directory and the .so is linked to a temporary name and renamed onto its final path, so
# concurrent builds (several ranks, two checkouts) never link each other's objects and nobody dlopens a half-written file.
set -e
cd "$(dirname "$0")"
OUT=../libgtav_amd.so
DEFS=""
if [ "$1" = "exp" ]; then
  shift
  OUT=../libgtav_amd_exp.so
  DEFS="-DGTAV_EXPERIMENTS"
fi
BUILD=$(mktemp -d "${TMPDIR:-/tmp}/gtav_build.XXXXXX")
trap 'rm -rf "$BUILD"' EXIT
pids=()
for f in gemm skinny elementwise attention train comm api api_dit api_train api_vae; do
  extra=""
  # attention: keep MFMA accumulators in VGPRs — the softmax works on them every key block, and the AGPR form costs 56
  # v_accvgpr moves per 16 MFMAs there (the GEMMs touch their accumulators only in the epilogue and keep the default)
  [ $f = attention ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed $DEFS $extra "$@" -c $f.hip -o "$BUILD/$f.o" &
  pids+=($!)
done
# the bf16-operand twins (common.h "operand type", ops_bf16.h): the same three sources with f16 = __bf16 in a namespace of their own.  Never with -DGTAV_EXPERIMENTS:
# the laboratory kernels exist for fp16 only.
for f in gemm elementwise attention; do
  extra=""
  [ $f = attention ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed -DGTAV_BF16_OPERANDS -Dgtav=gtav_bf16 $extra "$@" -c $f.hip -o "$BUILD/${f}_bf16.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$BUILD/lib.so" "$BUILD"/{gemm,skinny,elementwise,attention,train,comm,api,api_dit,api_train,api_vae,gemm_bf16,elementwise_bf16,attention_bf16}.o -ldl
mv -f "$BUILD/lib.so" "$OUT.tmp.$$"
mv -f "$OUT.tmp.$$" "$OUT"
echo "built $(realpath $OUT)"
