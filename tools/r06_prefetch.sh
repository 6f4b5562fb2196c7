#!/bin/bash
# A/B: register sets of the one-wave forward's record pipeline (GS_FWD_PREFETCH 1 / 2 / 3)
out=gpurun_out/r06_prefetch; rm -rf $out; mkdir -p $out
for rep in 1 2; do
for v in _pf1 "" _pf3; do
  GSPLAT_LIB=$PWD/gaussiansplattingmlx_amd/libgsplat_hip$v.so timeout -k 10 200 python bench.py --steps 100 --warmup 30 --no-cpu-baseline > $out/c3${v}_$rep.json 2>$out/err.txt || exit 1
done; done
for v in _pf1 "" _pf3; do
  GSPLAT_LIB=$PWD/gaussiansplattingmlx_amd/libgsplat_hip$v.so timeout -k 10 300 python bench.py --config c5_garden_2m --steps 100 --no-cpu-baseline > $out/c5${v}.json 2>>$out/err.txt || exit 1
  GSPLAT_LIB=$PWD/gaussiansplattingmlx_amd/libgsplat_hip$v.so timeout -k 10 300 python bench.py --config c2_100k_800 --steps 100 --no-cpu-baseline > $out/c2${v}.json 2>>$out/err.txt || exit 1
  GSPLAT_FWD_PAIR=0 GSPLAT_LIB=$PWD/gaussiansplattingmlx_amd/libgsplat_hip$v.so timeout -k 10 300 python bench.py --config c3_grown_1m --steps 100 --no-cpu-baseline > $out/grown_onewave${v}.json 2>>$out/err.txt || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_prefetch/*.json')):
    j = json.load(open(f))
    print(f.split('/')[-1], j['value'], j['ms_per_step'], {k: v['ms'] for k, v in j['stages'].items()})
PY
