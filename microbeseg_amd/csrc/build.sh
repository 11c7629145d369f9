#!/bin/bash
# Builds libmseg_hip.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot).
set -e
cd "$(dirname "$0")"
OUT=../libmseg_hip.so
SRCS="igemm.hip igemm_p8.hip wgrad.hip first.hip norm.hip head.hip loss.hip augment.hip labels.hip polygons.hip api_misc.hip"
[ -f postproc.hip ] && SRCS="$SRCS postproc.hip"
mkdir -p ../_build
OBJS=""
pids=()
for s in $SRCS; do
  o=../_build/${s%.hip}.o
  OBJS="$OBJS $o"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ common.h -nt "$o" ] || [ igemm_common.h -nt "$o" ] || [ bf16_affine.h -nt "$o" ] || [ ../../include/mseg_hip.h -nt "$o" ]; then
    # -pragma-unroll-threshold: the fully unrolled epilogues exceed LLVM's default 16k-instruction cap for "#pragma unroll";
    # a loop left rolled would index the accumulator array dynamically and push it to scratch memory
    # -fno-slp-vectorize for the matrix kernels: the SLP vectorizer turns the staging code's scalar fp32 multiply-adds into
    # v_pk_fma_f32 / v_pk_mul_f32, which beside MFMAs cost more than the two scalar instructions they replace
    # (MI355X_MICROARCH.md, cycle constants; measured here: bf16 320^2 step 38.13 -> 37.68 ms, fp32 neutral)
    extra=""
    case "$s" in igemm.hip|igemm_p8.hip|wgrad.hip) extra="-fno-slp-vectorize";; esac
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=200000 $extra -c "$s" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJS
echo "built $OUT"
