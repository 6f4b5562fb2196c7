#!/bin/bash
out=gpurun_out/resid; rm -rf $out; mkdir -p $out
for r in 4,16 5,16 6,16 3,16 4,12 4,16 6,16; do
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --residency $r > $out/r_$r.json 2>/dev/null
  python - $out/r_$r.json $r <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); print(sys.argv[2], j['value'], j['ms_per_step'], {k:v['ms'] for k,v in j['stages'].items() if k.startswith('blend')})
PY
done
