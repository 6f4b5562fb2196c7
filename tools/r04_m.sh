#!/bin/bash
for i in 1 2; do
for v in 100 8; do
python bench.py --steps 20 --warmup 5 --views $v --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('views $v steps 20:', j['value'], j['ms_per_step'], j['step_ms_spread'])"
done
done
python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('default flags (50/10):', j['value'], j['ms_per_step'])"
