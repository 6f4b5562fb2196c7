#!/bin/bash
# round-6 bench lines (one box, after the profile summaries are in profiles/): every config, the driver's 20-step line, the data-parallel
# rehearsals, config 4's arithmetic on one card
out=gpurun_out/r06_lines; rm -rf $out; mkdir -p $out
run() { name=$1; shift; timeout -k 10 500 python bench.py "$@" > $out/$name.json 2> $out/$name.err && echo "$name ok" || { echo "$name FAILED"; tail -3 $out/$name.err; }; }
run r06_bench_default --steps 100 --warmup 10
run r06_bench_c3_20steps --steps 20 --warmup 5 --no-cpu-baseline
run r06_bench_driver_defaults
run r06_bench_c1_10k_400_forward --config c1_10k_400 --steps 200 --warmup 20
run r06_bench_c2_100k_800_fwdbwd --config c2_100k_800 --steps 100 --warmup 10
run r06_bench_c5_garden_2m_240steps --config c5_garden_2m --steps 240 --warmup 10 --no-cpu-baseline
run r06_bench_c3_grown_1m_190steps --config c3_grown_1m --steps 190 --warmup 10 --no-cpu-baseline
GSPLAT_FWD_PAIR=0 run r06_bench_c3_grown_1m_190steps_one_wave_forward --config c3_grown_1m --steps 190 --warmup 10 --no-cpu-baseline
run r06_bench_tile200_block_lists_100steps --tile 200 --steps 100 --warmup 10 --no-cpu-baseline
GSPLAT_FWD_SLOW_SLOT=16 run r06_bench_c3_without_the_slow_slot_rule --steps 100 --warmup 10 --no-cpu-baseline
run r06_bench_single --steps 60 --warmup 10 --no-cpu-baseline
for impl in native torch; do
  run r06_bench_dp1_${impl} --steps 60 --warmup 10 --dp-single --dp-impl $impl --no-cpu-baseline
  run r06_bench_dp1_${impl}_20steps --steps 20 --warmup 5 --dp-single --dp-impl $impl --no-cpu-baseline
done
GSPLAT_DP_INLINE_GATHER=0 run r06_bench_dp1_native_side_stream_gather --steps 60 --warmup 10 --dp-single --dp-impl native --no-cpu-baseline
GSPLAT_PLANNED_DENSIFY=0 run r06_bench_dp1_native_20steps_unplanned_densify --steps 20 --warmup 5 --dp-single --dp-impl native --no-cpu-baseline
run r06_bench_dp1_native_allreduce --steps 60 --warmup 10 --dp-single --dp-impl native --dp-exchange allreduce --no-cpu-baseline
run r06_bench_c4_local8 --steps 24 --warmup 4 --views-per-step 8 --no-cpu-baseline
run r06_bench_c4_local8_dp1_torch --steps 24 --warmup 4 --views-per-step 8 --dp-single --dp-impl torch --no-cpu-baseline
GSPLAT_BENCH_DEVICE=0 timeout -k 10 400 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-cpu-baseline > $out/r06_bench_dp2_gloo_one_card.json 2> $out/r06_bench_dp2_gloo_one_card.err && echo "dp2 gloo ok" || tail -5 $out/r06_bench_dp2_gloo_one_card.err
GSPLAT_BENCH_DEVICE=0 timeout -k 10 500 python bench.py --gpus 4 --backend gloo --steps 12 --warmup 4 --no-cpu-baseline > $out/r06_bench_dp4_gloo_one_card.json 2> $out/r06_bench_dp4_gloo_one_card.err && echo "dp4 gloo ok" || tail -5 $out/r06_bench_dp4_gloo_one_card.err
timeout -k 10 300 python tools/dp_overflow_rehearsal.py > $out/r06_dp_overflow_rehearsal_2ranks_gloo_one_card.txt 2>&1; echo "rehearsal rc=$?"; tail -2 $out/r06_dp_overflow_rehearsal_2ranks_gloo_one_card.txt
timeout -k 10 300 python tools/dp_overflow_rehearsal.py one_view > $out/r06_dp_overflow_rehearsal_one_view_2ranks_gloo_one_card.txt 2>&1; echo "rehearsal one_view rc=$?"; tail -2 $out/r06_dp_overflow_rehearsal_one_view_2ranks_gloo_one_card.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_lines/*.json')):
    try: j = json.load(open(f))
    except Exception as e: print(f, 'BAD', e); continue
    r = j['roofline']
    print(f.split('/')[-1], j['value'], j['unit'], j['ms_per_step'], r['kernel'], r['frac'], r.get('frac_by_counters'), r.get('frac_claimed'), 'traffic', r.get('traffic_over_algorithmic'), r.get('issue_model_frac'), {k: v['ms'] for k, v in j['stages'].items()}, j['accounting_violations'], j.get('replicas_identical'))
    if j.get('exchange'): print('    exchange', {k: v for k, v in j['exchange'].items() if k.endswith('_ms') or k.startswith('coll')})
PY
