#!/bin/bash
python -m pytest tests/test_gpu_parity.py -x -q -k "checkpoint_arena or overflow" 2>&1 | tail -2
for v in "8 1" "8 0" "1 0" "8 1" "8 0" "1 0"; do set -- $v
GSPLAT_FWD_QUEUES=$1 GSPLAT_FWD_SPATIAL=$2 python bench.py --config c3_grown_1m --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('grown queues $1 spatial $2', j['ms_per_step'], 'fwd', j['stages']['blend_fwd']['ms'], 'bwd', j['stages']['blend_bwd']['ms'])"
done
for v in "8 1" "8 0" "1 0" "8 1" "8 0" "1 0"; do set -- $v
GSPLAT_FWD_QUEUES=$1 GSPLAT_FWD_SPATIAL=$2 python tools/bwd_ab.py c3_300k_800 2>/dev/null | tail -1
done
