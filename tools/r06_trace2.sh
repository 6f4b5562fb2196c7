#!/bin/bash
out=gpurun_out/r06_trace2; rm -rf $out; mkdir -p $out
timeout -k 10 300 python tools/fwd_trace.py 2>&1 | grep -v amdgpu.ids > $out/fwd_trace_c3.txt; grep -A12 "by hardware wave slot" $out/fwd_trace_c3.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "config1 or config2 or bench_workload or four_wave or staging_wave or fused_render_forward or blend_deep or randomized or adversarial" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -12 $out/pytest.log
cp gpurun_out/gradient_elementwise_c1_10k_400_sh1.0_fw*.json $out/ 2>/dev/null
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $out/$name.json 2> $out/$name.err && echo "$name ok" || { echo "$name FAILED"; tail -8 $out/$name.err; }; }
run c3 --steps 60 --warmup 10 --no-cpu-baseline
run c1 --config c1_10k_400 --steps 200 --warmup 20 --no-cpu-baseline
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_trace2/c*.json')):
    j = json.load(open(f)); print(f.split('/')[-1], j['value'], j['unit'], j['ms_per_step'], {k: v['ms'] for k, v in j['stages'].items()})
for f in sorted(glob.glob('gpurun_out/r06_trace2/gradient_elementwise_c1*.json')):
    j = json.load(open(f)); print(f.split('/')[-1], {k: (round(v['hip_vs_oracle32'], 4), round(v['oracle32_vs_oracle64'], 4), '%.1e' % v['max_norm_rel']) for k, v in j['tensors'].items()})
PY
