#!/bin/bash
# A/B of GS_TUNE_TRIM_RECTS on the bench scene and the grown scene (one box)
out=gpurun_out/r06_trim; mkdir -p $out
for t in 1 0 1 0; do
  GSPLAT_TRIM_RECTS=$t timeout -k 10 200 python bench.py --steps 100 --warmup 30 --no-cpu-baseline > $out/c3_trim${t}_$RANDOM.json 2>$out/err.txt || exit 1
done
for t in 1 0; do
  GSPLAT_TRIM_RECTS=$t timeout -k 10 300 python bench.py --config c3_grown_1m --steps 190 --no-cpu-baseline > $out/grown_trim${t}.json 2>>$out/err.txt || exit 1
  GSPLAT_TRIM_RECTS=$t timeout -k 10 300 python bench.py --config c5_garden_2m --steps 100 --no-cpu-baseline > $out/c5_trim${t}.json 2>>$out/err.txt || exit 1
done
echo done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_trim/*.json')):
    j = json.load(open(f))
    print(f.split('/')[-1], j['value'], j['ms_per_step'], {k: v['ms'] for k, v in j['stages'].items()}, j.get('M'), j['config'].get('pairs'))
PY
