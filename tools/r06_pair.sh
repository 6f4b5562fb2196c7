#!/bin/bash
# round 6, item 4: the staging-wave forward (GS_TUNE_FWD_PAIR) -- same-bits test, then A/B of the forward on every config; item 3: inline gather
out=gpurun_out/r06_pair; rm -rf $out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectory.py -m gpu -q -x -k "staging_wave or native or exchange or rccl" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -5 $out/pytest.log
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $out/$name.json 2> $out/$name.err && echo "$name ok" || { echo "$name FAILED"; tail -8 $out/$name.err; }; }
for pair in 0 12 14 10 8; do
  GSPLAT_FWD_PAIR=$pair run c3_pair$pair --steps 60 --warmup 10 --no-cpu-baseline
done
for pair in 0 12 14; do
  GSPLAT_FWD_PAIR=$pair run c2_pair$pair --config c2_100k_800 --steps 60 --warmup 10 --no-cpu-baseline
  GSPLAT_FWD_PAIR=$pair run grown_pair$pair --config c3_grown_1m --steps 60 --warmup 10 --no-cpu-baseline
  GSPLAT_FWD_PAIR=$pair run c5_pair$pair --config c5_garden_2m --steps 60 --warmup 10 --no-cpu-baseline
  GSPLAT_FWD_PAIR=$pair run t200_pair$pair --tile 200 --steps 40 --warmup 5 --no-cpu-baseline
done
run dp1_native --steps 60 --warmup 10 --dp-single --dp-impl native --no-cpu-baseline
GSPLAT_DP_INLINE_GATHER=0 run dp1_native_side_gather --steps 60 --warmup 10 --dp-single --dp-impl native --no-cpu-baseline
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_pair/*.json')):
    try: j = json.load(open(f))
    except Exception as e: print(f, 'BAD', e); continue
    print(f.split('/')[-1], j['value'], j['unit'], j['ms_per_step'], 'fwd_ms', j['fwd_ms'], {k: v['ms'] for k, v in j['stages'].items()})
    if j.get('exchange'): print('    exchange', {k: v for k, v in j['exchange'].items() if k.endswith('_ms') or k.startswith('coll')})
PY
