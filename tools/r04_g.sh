#!/bin/bash
out=gpurun_out/r04g; mkdir -p $out
for v in "1 1" "3 1" "1 0" ; do
  set -- $v
  GSPLAT_COLOUR_RIDERS=$1 GSPLAT_SS_BIG=$2 python bench.py --config c3_grown_1m --steps 60 --warmup 10 --no-cpu-baseline > $out/grown_r$1_b$2.json 2> $out/err.txt && echo "grown riders=$1 big=$2 ok"
  GSPLAT_COLOUR_RIDERS=$1 GSPLAT_SS_BIG=$2 python bench.py --config c5_garden_2m --steps 60 --warmup 10 --views 8 --no-cpu-baseline > $out/c5_r$1_b$2.json 2> $out/err.txt && echo "c5 riders=$1 big=$2 ok"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04g/*.json')):
    j = json.load(open(f))
    print(f.split('/')[-1], j['ms_per_step'], {k: v['ms'] for k, v in j['stages'].items()})
PY
