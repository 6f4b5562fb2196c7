#!/bin/bash
out=gpurun_out/r04c; mkdir -p $out
for v in default q40 q66; do
  lib=""; [ $v != default ] && lib=$PWD/gaussiansplattingmlx_amd/libgsplat_hip_$v.so
  GSPLAT_LIB=$lib python tools/bwd_ab.py c3_300k_800 > $out/bwd_$v.json 2>/dev/null; cat $out/bwd_$v.json
  GSPLAT_LIB=$lib python tools/bwd_ab.py c3_grown_1m > $out/bwdg_$v.json 2>/dev/null; cat $out/bwdg_$v.json
done
python -m pytest tests/test_gpu_trajectory.py -q > $out/traj.log 2>&1; tail -3 $out/traj.log
mkdir -p $out/default; mv gpurun_out/trajectory_*.json $out/default/
GSPLAT_LIB=$PWD/gaussiansplattingmlx_amd/libgsplat_hip_q40.so python -m pytest tests/test_gpu_trajectory.py -q > $out/traj_q40.log 2>&1; tail -3 $out/traj_q40.log
mkdir -p $out/q40; mv gpurun_out/trajectory_*.json $out/q40/
python bench.py --config c3_grown_1m --steps 60 --warmup 10 --no-cpu-baseline > $out/bench_grown.json 2> $out/bench_grown.err && echo grown ok
