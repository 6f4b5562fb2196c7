#!/bin/bash
# the staging-wave forward against the one-wave forward on TRIMMED lists (GS_TUNE_TRIM_RECTS = 1): grown scene, c5, tile 200
out=gpurun_out/r06_pairpolicy; rm -rf $out; mkdir -p $out
for fp in -1 0 1; do
  GSPLAT_FWD_PAIR=$fp timeout -k 10 300 python bench.py --config c3_grown_1m --steps 100 --no-cpu-baseline > $out/grown_fp$fp.json 2>>$out/err.txt || exit 1
  GSPLAT_FWD_PAIR=$fp timeout -k 10 300 python bench.py --config c5_garden_2m --steps 60 --no-cpu-baseline > $out/c5_fp$fp.json 2>>$out/err.txt || exit 1
  GSPLAT_FWD_PAIR=$fp timeout -k 10 300 python bench.py --tile 200 --steps 60 --no-cpu-baseline > $out/t200_fp$fp.json 2>>$out/err.txt || exit 1
  GSPLAT_TRIM_RECTS=0 GSPLAT_FWD_PAIR=$fp timeout -k 10 300 python bench.py --config c3_grown_1m --steps 100 --no-cpu-baseline > $out/grown_untrimmed_fp$fp.json 2>>$out/err.txt || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_pairpolicy/*.json')):
    j = json.load(open(f))
    print(f.split('/')[-1], j['value'], j['ms_per_step'], {k: v['ms'] for k, v in j['stages'].items()})
PY
