#!/bin/bash
# round 5: arithmetic masks in place of the half-rate compares / selects of the blend inner loops (EXPERIMENTS.md): fixed scene,
# stage time per library build, alternating on one box; the parity tests under each variant build
out=gpurun_out/r05_mask_ab; rm -rf $out; mkdir -p $out
L=gaussiansplattingmlx_amd
for rep in 1 2 3; do
  for v in "" _fwdmask _bwdmask; do
    GSPLAT_LIB=$PWD/$L/libgsplat_hip$v.so python tools/bwd_ab.py c3_300k_800 >> $out/ab_c3.txt 2>> $out/ab.err
  done
done
for v in "" _fwdmask _bwdmask; do GSPLAT_LIB=$PWD/$L/libgsplat_hip$v.so python tools/bwd_ab.py c2_100k_800 >> $out/ab_c2.txt 2>> $out/ab.err; done
cat $out/ab_c3.txt $out/ab_c2.txt
for v in _fwdmask _bwdmask; do
  GSPLAT_LIB=$PWD/$L/libgsplat_hip$v.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_render or randomized or config2 or bench_workload or adversarial or depth_cuts_are_exact" > $out/pytest$v.log 2>&1; echo "variant $v pytest rc=$?"; tail -3 $out/pytest$v.log
done
