#!/bin/bash
# round-5 bench lines (one box): every config, the driver's 20-step line, the reference app's tile size, the densify A/B
out=gpurun_out/r05_lines; rm -rf $out; mkdir -p $out
run() { name=$1; shift; python bench.py "$@" > $out/$name.json 2> $out/$name.err && echo "$name ok" || { echo "$name FAILED"; tail -3 $out/$name.err; }; }
run r05_bench_default --steps 100 --warmup 10
run r05_bench_c3_20steps --steps 20 --warmup 5 --no-cpu-baseline
GSPLAT_PLANNED_DENSIFY=0 run r05_bench_c3_20steps_unplanned_densify --steps 20 --warmup 5 --no-cpu-baseline
run r05_bench_c1_10k_400_forward --config c1_10k_400 --steps 200 --warmup 20
run r05_bench_c2_100k_800_fwdbwd --config c2_100k_800 --steps 100 --warmup 10
run r05_bench_c5_garden_2m_240steps --config c5_garden_2m --steps 240 --warmup 10 --no-cpu-baseline
run r05_bench_c3_grown_1m_190steps --config c3_grown_1m --steps 190 --warmup 10 --no-cpu-baseline
run r05_bench_tile200_block_lists --tile 200 --steps 40 --warmup 5 --no-cpu-baseline
run r05_bench_tile200_block_lists_100steps --tile 200 --steps 100 --warmup 10 --no-cpu-baseline
GSPLAT_BLOCK_LISTS=0 run r05_bench_tile200_generic_kernels --tile 200 --steps 40 --warmup 5 --no-cpu-baseline
GSPLAT_CUT_SUPER=0 run r05_bench_c5_garden_2m_240steps_without_coarse_cuts --config c5_garden_2m --steps 240 --warmup 10 --no-cpu-baseline
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05_lines/*.json')):
    try: j = json.load(open(f))
    except Exception as e: print(f, 'BAD', e); continue
    r = j['roofline']
    print(f.split('/')[-1], j['value'], j['unit'], j['ms_per_step'], r['kernel'], r['frac'], 'traffic', r['traffic'], r.get('traffic_over_algorithmic'), r.get('issue_model_frac'), {k: v['ms'] for k, v in j['stages'].items()}, j['accounting_violations'])
PY
