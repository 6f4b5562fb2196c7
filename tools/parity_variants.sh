#!/bin/bash
# The image bar per forward-arithmetic variant: run on the GPU box through gpurun.
out=$GRAFT_REPO_ROOT/gpurun_out/parity_variants; mkdir -p $out
for v in "" _unfused _libm _unfused_libm; do
  GSPLAT_LIB=$GRAFT_REPO_ROOT/gaussiansplattingmlx_amd/libgsplat_hip$v.so python3 tools/full_size_parity.py > $out/parity${v:-_default}.json 2> $out/parity${v:-_default}.err || exit 1
done
