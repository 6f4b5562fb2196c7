#!/bin/bash
# bench line (train step) per setting of the rider shares, one box
for sh in "350,450,200" "300,400,150" "250,350,150" "300,450,250" "400,450,150" "350,400,250" "350,450,200"; do
  GSPLAT_RIDER_SHARES=$sh python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('shares $sh', d['value'], d['ms_per_step'], d['stages']['proj_fwd']['ms'], d['stages']['bin']['ms'])"
done
