#!/bin/bash
# the bench line per setting of GS_TUNE_COLOUR_RIDERS (0 interleaved kernel, 3 geometry-then-colours, 1 riders), alternating, one box
for i in 1 2; do for m in 0 3 1; do
  GSPLAT_COLOUR_RIDERS=$m python bench.py --no-cpu-baseline --steps 100 --warmup 10 "$@" 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('colour_riders $m', d['value'], d['ms_per_step'], d['stages']['proj_fwd']['ms'], d['stages']['bin']['ms'])"
done; done
