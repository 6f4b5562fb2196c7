#!/bin/bash
out=gpurun_out/r06_trace; rm -rf $out; mkdir -p $out
timeout -k 10 300 python tools/fwd_trace.py 2>&1 | grep -v amdgpu.ids > $out/fwd_trace_c3.txt; tail -32 $out/fwd_trace_c3.txt
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $out/$name.json 2> $out/$name.err && echo "$name ok" || { echo "$name FAILED"; tail -8 $out/$name.err; }; }
run grown_res6 --config c3_grown_1m --steps 60 --warmup 10 --no-cpu-baseline --residency 6,16
run grown_res5 --config c3_grown_1m --steps 60 --warmup 10 --no-cpu-baseline --residency 5,16
run grown_res4 --config c3_grown_1m --steps 60 --warmup 10 --no-cpu-baseline
GSPLAT_FWD_PAIR=12 run grown_pair12 --config c3_grown_1m --steps 60 --warmup 10 --no-cpu-baseline
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "projection or tile_bin_bit or fused_render_forward or config1 or bench_workload" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -4 $out/pytest.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_trace/*.json')):
    try: j = json.load(open(f))
    except Exception as e: print(f, 'BAD', e); continue
    print(f.split('/')[-1], j['value'], j['unit'], j['ms_per_step'], 'fwd_ms', j['fwd_ms'], {k: v['ms'] for k, v in j['stages'].items()})
PY
