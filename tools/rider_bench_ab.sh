#!/bin/bash
# the bench line with and without the colour riders, alternating, one box
for i in 1 2; do for m in 0 1; do
  GSPLAT_COLOUR_RIDERS=$m python bench.py --no-cpu-baseline --steps 100 --warmup 10 "$@" 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('riders $m', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()})"
done; done
