#!/bin/bash
p() { python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$1', j['value'], j['ms_per_step'], j['densify']['at_iterations'], j['densify']['N_after'], j['step_ms_spread'])"; }
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | p "event in region      "
GSPLAT_BENCH_IT0=601 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | p "no event, N=300k     "
GSPLAT_BENCH_IT0=580 python bench.py --steps 20 --warmup 25 --no-cpu-baseline 2>/dev/null | p "event in warm-up     "
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | p "event in region      "
