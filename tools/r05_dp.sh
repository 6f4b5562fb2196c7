#!/bin/bash
# round-5 data-parallel checks on one box: the DP tests, the 1-rank rehearsals of both issuers / exchanges, the gloo rehearsals
out=gpurun_out/r05_dp; rm -rf $out; mkdir -p $out
python -m pytest tests -m gpu -x -q -k "exchange or trajectory or replica or rccl" > $out/pytest_dp.log 2>&1; echo "pytest rc=$?" >> $out/pytest_dp.log; tail -4 $out/pytest_dp.log
run() { name=$1; shift; python bench.py "$@" > $out/$name.json 2> $out/$name.err && echo "$name ok" || { echo "$name FAILED"; tail -5 $out/$name.err; }; }
run r05_bench_dp1_torch --steps 60 --warmup 10 --dp-single --dp-impl torch --no-cpu-baseline
run r05_bench_dp1_native --steps 60 --warmup 10 --dp-single --dp-impl native --no-cpu-baseline
run r05_bench_dp1_native_allreduce --steps 60 --warmup 10 --dp-single --dp-impl native --dp-exchange allreduce --no-cpu-baseline
run r05_bench_dp1_torch_allreduce --steps 60 --warmup 10 --dp-single --dp-impl torch --dp-exchange allreduce --no-cpu-baseline
run r05_bench_single --steps 60 --warmup 10 --no-cpu-baseline
GSPLAT_BENCH_DEVICE=0 timeout -k 10 400 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-cpu-baseline > $out/r05_bench_dp2_gloo_one_card.json 2> $out/r05_bench_dp2_gloo_one_card.err && echo "dp2 gloo ok" || tail -5 $out/r05_bench_dp2_gloo_one_card.err
timeout -k 10 300 python tools/dp_overflow_rehearsal.py > $out/r05_dp_overflow_rehearsal_2ranks_gloo_one_card.txt 2>&1; echo "rehearsal rc=$?"; tail -3 $out/r05_dp_overflow_rehearsal_2ranks_gloo_one_card.txt
timeout -k 10 300 python tools/dp_overflow_rehearsal.py one_view > $out/r05_dp_overflow_rehearsal_one_view_2ranks_gloo_one_card.txt 2>&1; echo "rehearsal one_view rc=$?"; tail -3 $out/r05_dp_overflow_rehearsal_one_view_2ranks_gloo_one_card.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05_dp/*.json')):
    try: j = json.load(open(f))
    except Exception as e: print(f, 'BAD', e); continue
    print(f.split('/')[-1], j['value'], j['unit'], j['ms_per_step'], {k: v['ms'] for k, v in j['stages'].items()}, j.get('accounting_violations'))
    if j.get('exchange'): print('    exchange', {k: v for k, v in j['exchange'].items() if k.endswith('_ms') or k.startswith('coll')}, j['replicas_identical'])
PY
