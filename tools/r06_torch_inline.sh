#!/bin/bash
# torch issuer: the all-gather as a synchronous op on the render stream (default) against ProcessGroupNCCL's own stream
out=gpurun_out/r06_torch_inline; rm -rf $out; mkdir -p $out
run() { name=$1; shift; timeout -k 10 300 python bench.py "$@" > $out/$name.json 2> $out/$name.err && echo "$name ok" || { echo "$name FAILED"; tail -5 $out/$name.err; exit 1; }; }
for rep in 1 2; do
run dp1_torch_inline_$rep --steps 60 --warmup 10 --dp-single --dp-impl torch --no-cpu-baseline
GSPLAT_DP_INLINE_GATHER=0 run dp1_torch_own_stream_$rep --steps 60 --warmup 10 --dp-single --dp-impl torch --no-cpu-baseline
done
run dp1_native --steps 60 --warmup 10 --dp-single --dp-impl native --no-cpu-baseline
run single --steps 60 --warmup 10 --no-cpu-baseline
run c4_local8_dp1_torch_inline --steps 24 --warmup 4 --views-per-step 8 --dp-single --dp-impl torch --no-cpu-baseline
timeout -k 10 600 python -m pytest tests -m gpu -q -k "torch or dp1 or planned" -p no:cacheprovider > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_torch_inline/*.json')):
    j = json.load(open(f))
    print(f.split('/')[-1], j['value'], j['ms_per_step'], {k: v for k, v in (j.get('exchange') or {}).items() if k.endswith('_ms')})
PY
