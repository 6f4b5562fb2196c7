#!/bin/bash
# the torch-issuer lines again (the all-gather now a synchronous op on the render stream), with the single-device and native lines of the same box beside them
out=gpurun_out/r06_lines_torch; rm -rf $out; mkdir -p $out
run() { name=$1; shift; timeout -k 10 500 python bench.py "$@" > $out/$name.json 2> $out/$name.err && echo "$name ok" || { echo "$name FAILED"; tail -3 $out/$name.err; }; }
run r06_bench_dp1_torch --steps 60 --warmup 10 --dp-single --dp-impl torch --no-cpu-baseline
run r06_bench_dp1_torch_20steps --steps 20 --warmup 5 --dp-single --dp-impl torch --no-cpu-baseline
GSPLAT_DP_INLINE_GATHER=0 run r06_bench_dp1_torch_gather_on_its_own_stream --steps 60 --warmup 10 --dp-single --dp-impl torch --no-cpu-baseline
run r06_bench_c4_local8_dp1_torch --steps 24 --warmup 4 --views-per-step 8 --dp-single --dp-impl torch --no-cpu-baseline
run same_box_single --steps 60 --warmup 10 --no-cpu-baseline
run same_box_dp1_native --steps 60 --warmup 10 --dp-single --dp-impl native --no-cpu-baseline
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_lines_torch/*.json')):
    j = json.load(open(f))
    print(f.split('/')[-1], j['value'], j['ms_per_step'], j['accounting_violations'], {k: v for k, v in (j.get('exchange') or {}).items() if k.endswith('_ms')})
PY
