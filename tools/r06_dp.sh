#!/bin/bash
# round 6, items 1-2: the eight-view step and the planned densify event of the data-parallel form -- tests, then 20- and 100-step lines
out=gpurun_out/r06_dp; rm -rf $out; mkdir -p $out
timeout -k 10 1100 python -m pytest tests/test_gpu_eight_views.py tests/test_gpu_trajectory.py tests/test_gpu_parity.py -m gpu -x -q -k "eight or local8 or planned or exchange or replica or rccl or densify or split_and_prune" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -15 $out/pytest.log
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $out/$name.json 2> $out/$name.err && echo "$name ok" || { echo "$name FAILED"; tail -8 $out/$name.err; }; }
run r06_bench_single_100 --steps 100 --warmup 10 --no-cpu-baseline
run r06_bench_single_20 --steps 20 --warmup 5 --no-cpu-baseline
for impl in native torch; do
  run r06_bench_dp1_${impl}_100 --steps 100 --warmup 10 --dp-single --dp-impl $impl --no-cpu-baseline
  run r06_bench_dp1_${impl}_20 --steps 20 --warmup 5 --dp-single --dp-impl $impl --no-cpu-baseline
  GSPLAT_PLANNED_DENSIFY=0 run r06_bench_dp1_${impl}_20_unplanned --steps 20 --warmup 5 --dp-single --dp-impl $impl --no-cpu-baseline
done
run r06_bench_c4_local8 --steps 24 --warmup 4 --views-per-step 8 --no-cpu-baseline
run r06_bench_c4_local8_dp1_torch --steps 24 --warmup 4 --views-per-step 8 --dp-single --dp-impl torch --no-cpu-baseline
GSPLAT_BENCH_DEVICE=0 timeout -k 10 400 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-cpu-baseline > $out/r06_bench_dp2_gloo_one_card.json 2> $out/r06_bench_dp2_gloo_one_card.err && echo "dp2 gloo ok" || tail -8 $out/r06_bench_dp2_gloo_one_card.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_dp/*.json')):
    try: j = json.load(open(f))
    except Exception as e: print(f, 'BAD', e); continue
    print(f.split('/')[-1], j['value'], j['unit'], j['ms_per_step'], {k: v['ms'] for k, v in j['stages'].items()}, j.get('accounting_violations'), j['densify'])
    if j.get('exchange'): print('    exchange', {k: v for k, v in j['exchange'].items() if k.endswith('_ms') or k.startswith('coll')}, j['replicas_identical'])
PY
