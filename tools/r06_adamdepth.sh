#!/bin/bash
# A/B: batches of Adam moments in flight ahead of the update in the fused projection backward (GS_ADAM_ROWS_DEPTH 2 / 3 / 4)
out=gpurun_out/r06_adamdepth; mkdir -p $out
for rep in 1 2; do
for v in "" _d3 _d4; do
  GSPLAT_LIB=$PWD/gaussiansplattingmlx_amd/libgsplat_hip$v.so timeout -k 10 200 python bench.py --steps 100 --warmup 30 --no-cpu-baseline > $out/c3${v}_$rep.json 2>$out/err.txt || exit 1
done; done
for v in "" _d3 _d4; do
  GSPLAT_LIB=$PWD/gaussiansplattingmlx_amd/libgsplat_hip$v.so timeout -k 10 300 python bench.py --config c5_garden_2m --steps 100 --no-cpu-baseline > $out/c5${v}.json 2>>$out/err.txt || exit 1
  GSPLAT_LIB=$PWD/gaussiansplattingmlx_amd/libgsplat_hip$v.so timeout -k 10 300 python bench.py --steps 60 --dp-single --dp-impl native --no-cpu-baseline > $out/dp1${v}.json 2>>$out/err.txt || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_adamdepth/*.json')):
    j = json.load(open(f))
    print(f.split('/')[-1], j['value'], j['ms_per_step'], {k: v['ms'] for k, v in j['stages'].items()})
PY
