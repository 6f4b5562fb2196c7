#!/bin/bash
out=gpurun_out/r06_res; rm -rf $out; mkdir -p $out
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $out/$name.json 2> $out/$name.err && echo "$name ok" || { echo "$name FAILED"; tail -8 $out/$name.err; }; }
for w in 3 4 5 6; do for ss in 3 16 2; do
  GSPLAT_FWD_SLOW_SLOT=$ss run c3_w${w}_s$ss --steps 60 --warmup 10 --no-cpu-baseline --residency $w,16
done; done
for w in 3 5; do
  run c2_w${w} --config c2_100k_800 --steps 60 --warmup 10 --no-cpu-baseline --residency $w,16
  run grown_w${w} --config c3_grown_1m --steps 60 --warmup 10 --no-cpu-baseline --residency $w,16
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_res/*.json')):
    try: j = json.load(open(f))
    except Exception as e: print(f, 'BAD', e); continue
    print(f.split('/')[-1], j['value'], j['ms_per_step'], 'fwd_ms', j['fwd_ms'], 'blend_fwd', j['stages']['blend_fwd']['ms'])
PY
