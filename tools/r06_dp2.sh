#!/bin/bash
# round 6, items 1-3: DP tests on the split form (SH rows read once), the event's cost per form, bench lines
out=gpurun_out/r06_dp2; rm -rf $out; mkdir -p $out
timeout -k 10 1100 python -m pytest tests/test_gpu_eight_views.py tests/test_gpu_trajectory.py tests/test_gpu_parity.py -m gpu -q -k "eight or local8 or native or planned or exchange or replica or rccl or densify or split_and_prune or sh_compressed or sh_grad" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -25 $out/pytest.log
for f in single dp1_native dp1_torch local8; do for p in 1 0; do timeout -k 10 200 python tools/dp_event_cost.py $f $p 2>&1 | grep -v amdgpu.ids | tee -a $out/event_cost.txt; done; done
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $out/$name.json 2> $out/$name.err && echo "$name ok" || { echo "$name FAILED"; tail -8 $out/$name.err; }; }
run r06_bench_single --steps 60 --warmup 10 --no-cpu-baseline
for impl in native torch; do
  run r06_bench_dp1_${impl} --steps 60 --warmup 10 --dp-single --dp-impl $impl --no-cpu-baseline
  run r06_bench_dp1_${impl}_20 --steps 20 --warmup 5 --dp-single --dp-impl $impl --no-cpu-baseline
done
run r06_bench_c4_local8 --steps 24 --warmup 4 --views-per-step 8 --no-cpu-baseline
run r06_bench_c4_local8_dp1_torch --steps 24 --warmup 4 --views-per-step 8 --dp-single --dp-impl torch --no-cpu-baseline
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_dp2/*.json')):
    try: j = json.load(open(f))
    except Exception as e: print(f, 'BAD', e); continue
    print(f.split('/')[-1], j['value'], j['unit'], j['ms_per_step'], {k: v['ms'] for k, v in j['stages'].items()}, j.get('accounting_violations'))
    if j.get('exchange'): print('    exchange', {k: v for k, v in j['exchange'].items() if k.endswith('_ms') or k.startswith('coll')}, j['replicas_identical'])
PY
