#!/bin/bash
# round 6, item 1: config 4's eight-view step on one card -- its tests, then the side lines beside the single-device / 1-rank DP lines of the same box
out=gpurun_out/r06_c4; rm -rf $out; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_eight_views.py tests/test_gpu_trajectory.py -m gpu -x -q -k "eight or local8" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -15 $out/pytest.log
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $out/$name.json 2> $out/$name.err && echo "$name ok" || { echo "$name FAILED"; tail -8 $out/$name.err; }; }
run r06_bench_single --steps 60 --warmup 10 --no-cpu-baseline
run r06_bench_dp1_native --steps 60 --warmup 10 --dp-single --dp-impl native --no-cpu-baseline
run r06_bench_dp1_torch --steps 60 --warmup 10 --dp-single --dp-impl torch --no-cpu-baseline
run r06_bench_c4_local8 --steps 24 --warmup 4 --views-per-step 8 --no-cpu-baseline
run r06_bench_c4_local8_dp1_torch --steps 24 --warmup 4 --views-per-step 8 --dp-single --dp-impl torch --no-cpu-baseline
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_c4/*.json')):
    try: j = json.load(open(f))
    except Exception as e: print(f, 'BAD', e); continue
    print(f.split('/')[-1], j['value'], j['unit'], j['ms_per_step'], {k: v['ms'] for k, v in j['stages'].items()}, j.get('accounting_violations'))
    if j.get('exchange'): print('    exchange', {k: v for k, v in j['exchange'].items() if k.endswith('_ms') or k.startswith('coll')}, j['replicas_identical'])
PY
