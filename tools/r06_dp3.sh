#!/bin/bash
# round 6, item 3: the merged geometry kernel -- DP tests, per-kernel times of the 1-rank native step, event cost, lines
out=gpurun_out/r06_dp3; rm -rf $out; mkdir -p $out
timeout -k 10 1100 python -m pytest tests/test_gpu_eight_views.py tests/test_gpu_trajectory.py tests/test_gpu_parity.py -m gpu -q -k "eight or local8 or native or planned or exchange or replica or rccl or densify or split_and_prune or sh_compressed or sh_grad" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -5 $out/pytest.log
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $out/$name.json 2> $out/$name.err && echo "$name ok" || { echo "$name FAILED"; tail -8 $out/$name.err; }; }
run r06_bench_single --steps 60 --warmup 10 --no-cpu-baseline
for impl in native torch; do
  run r06_bench_dp1_${impl} --steps 60 --warmup 10 --dp-single --dp-impl $impl --no-cpu-baseline
done
run r06_bench_c4_local8 --steps 24 --warmup 4 --views-per-step 8 --no-cpu-baseline
root=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/stats_dp1 -o s -- python3 $root/bench.py --steps 30 --warmup 5 --dp-single --dp-impl native --no-cpu-baseline > $root/$out/stats_dp1_bench.json 2> $root/$out/stats_dp1.log )
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r06_dp3/stats_dp1/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:28]:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
for f in single dp1_native dp1_torch; do for p in 1 0; do timeout -k 10 200 python tools/dp_event_cost.py $f $p 2>&1 | grep "planned=" | tee -a $out/event_cost.txt; done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_dp3/r06*.json')):
    try: j = json.load(open(f))
    except Exception as e: print(f, 'BAD', e); continue
    print(f.split('/')[-1], j['value'], j['unit'], j['ms_per_step'], {k: v['ms'] for k, v in j['stages'].items()}, j.get('accounting_violations'))
    if j.get('exchange'): print('    exchange', {k: v for k, v in j['exchange'].items() if k.endswith('_ms') or k.startswith('coll')}, j['replicas_identical'])
PY
