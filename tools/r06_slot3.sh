#!/bin/bash
out=gpurun_out/r06_slot3; rm -rf $out; mkdir -p $out
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $out/$name.json 2> $out/$name.err && echo "$name ok" || { echo "$name FAILED"; tail -8 $out/$name.err; }; }
for lim in 0 200 300 210 320 2200; do
  GSPLAT_FWD_SLOT_LIMITS=$lim run c3_lim$lim --steps 60 --warmup 10 --no-cpu-baseline
  GSPLAT_FWD_SLOT_LIMITS=$lim run c2_lim$lim --config c2_100k_800 --steps 60 --warmup 10 --no-cpu-baseline
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_slot3/*.json')):
    try: j = json.load(open(f))
    except Exception as e: print(f, 'BAD', e); continue
    print(f.split('/')[-1], j['value'], j['unit'], j['ms_per_step'], 'fwd_ms', j['fwd_ms'], {k: v['ms'] for k, v in j['stages'].items()})
PY
