#!/bin/bash
# Build libcrab_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../libcrab_hip.so
# -pragma-unroll-threshold: "#pragma unroll" is silently dropped above 16k instructions; a dropped unroll turns the
# compile-time accumulator indices of the GEMM epilogues into scratch accesses
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -mllvm -pragma-unroll-threshold=200000"
mkdir -p build
pids=()
for f in *.hip; do
  o=build/${f%.hip}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ common.h -nt "$o" ] || [ crab_internal.h -nt "$o" ] || [ gemm_epilogue.h -nt "$o" ] || [ build.sh -nt "$o" ] || [ ../../include/crab_hip.h -nt "$o" ]; then
    extra=""
    # attn.hip: keep the MFMA accumulators in VGPRs.  The flash kernels rescale O every K/V tile; with AGPR accumulators that is
    # 32 v_accvgpr_read + 32 v_mov around 32 multiplies per tile (CLIP shape 150 -> 126 us, decoder shape 217 -> 213 us)
    if [ "$f" = "attn.hip" ]; then extra="-mllvm -amdgpu-mfma-vgpr-form"; fi
    /opt/rocm/bin/hipcc $FLAGS $extra -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o "$OUT" -ldl
echo "built $(realpath $OUT)"
