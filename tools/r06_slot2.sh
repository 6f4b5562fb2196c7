#!/bin/bash
out=gpurun_out/r06_slot2; rm -rf $out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "queue_count or staging_wave or config1 or depth_cuts or fused_render or four_waves or checkpoint_arena or reserved_overflow or block_lists" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -4 $out/pytest.log
timeout -k 10 300 python tools/fwd_trace.py 2>&1 | grep -v amdgpu.ids > $out/fwd_trace_c3.txt; grep -A8 "by hardware wave slot" $out/fwd_trace_c3.txt; grep "SIMDs with items\|one CU" $out/fwd_trace_c3.txt
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $out/$name.json 2> $out/$name.err && echo "$name ok" || { echo "$name FAILED"; tail -8 $out/$name.err; }; }
for ss in 3 16 2; do
  GSPLAT_FWD_SLOW_SLOT=$ss run c3_slot$ss --steps 60 --warmup 10 --no-cpu-baseline
  GSPLAT_FWD_SLOW_SLOT=$ss run c2_slot$ss --config c2_100k_800 --steps 60 --warmup 10 --no-cpu-baseline
  GSPLAT_FWD_SLOW_SLOT=$ss run grown_slot$ss --config c3_grown_1m --steps 60 --warmup 10 --no-cpu-baseline
  GSPLAT_FWD_SLOW_SLOT=$ss run c5_slot$ss --config c5_garden_2m --steps 60 --warmup 10 --no-cpu-baseline
  GSPLAT_FWD_SLOW_SLOT=$ss run t200_slot$ss --tile 200 --steps 40 --warmup 5 --no-cpu-baseline
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_slot2/*.json')):
    try: j = json.load(open(f))
    except Exception as e: print(f, 'BAD', e); continue
    print(f.split('/')[-1], j['value'], j['unit'], j['ms_per_step'], 'fwd_ms', j['fwd_ms'], {k: v['ms'] for k, v in j['stages'].items()})
PY
