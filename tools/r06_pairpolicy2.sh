#!/bin/bash
# the staging-wave forward on row-group lists: forced on / off on the grown scene, c5, c3
out=gpurun_out/r06_pairpolicy2; rm -rf $out; mkdir -p $out
for fp in 0 1 -1; do
  GSPLAT_FWD_PAIR=$fp timeout -k 10 300 python bench.py --config c3_grown_1m --steps 190 --no-cpu-baseline > $out/grown_fp$fp.json 2>>$out/err.txt || exit 1
  GSPLAT_FWD_PAIR=$fp timeout -k 10 300 python bench.py --config c5_garden_2m --steps 60 --no-cpu-baseline > $out/c5_fp$fp.json 2>>$out/err.txt || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_pairpolicy2/*.json')):
    j = json.load(open(f))
    print(f.split('/')[-1], j['value'], j['ms_per_step'], {k: v['ms'] for k, v in j['stages'].items()}, (j.get('workload_stats') or {}).get('M_pairs'))
PY
