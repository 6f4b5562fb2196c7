#!/bin/bash
# round-4 bench lines (one box): every config, the view-count A/B, the reference app's tile size, the exchange rehearsals
out=gpurun_out/r04_lines; rm -rf $out; mkdir -p $out
run() { name=$1; shift; python bench.py "$@" > $out/$name.json 2> $out/$name.err && echo "$name ok" || { echo "$name FAILED"; tail -3 $out/$name.err; }; }
run r04_bench_default --steps 100 --warmup 10
run r04_bench_c3_views8 --steps 100 --warmup 10 --views 8 --no-cpu-baseline
run r04_bench_c3_20steps --steps 20 --warmup 5 --no-cpu-baseline
run r04_bench_c1_10k_400_forward --config c1_10k_400 --steps 200 --warmup 20
run r04_bench_c2_100k_800_fwdbwd --config c2_100k_800 --steps 100 --warmup 10
run r04_bench_c5_garden_2m_240steps --config c5_garden_2m --steps 240 --warmup 10 --no-cpu-baseline
run r04_bench_c3_grown_1m_190steps --config c3_grown_1m --steps 190 --warmup 10 --no-cpu-baseline
run r04_bench_tile200 --tile 200 --steps 40 --warmup 5 --no-cpu-baseline
run r04_bench_dp1_torch --steps 60 --warmup 10 --dp-single --dp-impl torch --no-cpu-baseline
run r04_bench_dp1_native --steps 60 --warmup 10 --dp-single --dp-impl native --no-cpu-baseline
run r04_bench_dp1_native_allreduce --steps 60 --warmup 10 --dp-single --dp-impl native --dp-exchange allreduce --no-cpu-baseline
GSPLAT_BENCH_DEVICE=0 timeout -k 10 400 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-cpu-baseline > $out/r04_bench_dp2_gloo_one_card.json 2> $out/r04_bench_dp2_gloo_one_card.err && echo "dp2 gloo ok"
timeout -k 10 120 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/r04_bench_dp2_refused_up_front.txt; echo "exit code $?" >> $out/r04_bench_dp2_refused_up_front.txt
GSPLAT_BENCH_DEVICE=0 timeout -k 10 180 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/nccl_one_card.err; echo "exit code $?" >> $out/nccl_one_card.err
grep -E "bench.py: the 2-rank|Duplicate GPU|ncclInvalidUsage|exit code" $out/nccl_one_card.err | sort | uniq -c | head -8 > $out/r04_bench_dp2_nccl_one_card_refused.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04_lines/*.json')):
    try: j = json.load(open(f))
    except Exception as e: print(f, 'BAD', e); continue
    print(f.split('/')[-1], j['value'], j['unit'], j['ms_per_step'], j['roofline']['kernel'], j['roofline']['frac'], 'traffic', j['roofline']['traffic'], {k: v['ms'] for k, v in j['stages'].items()})
    if j.get('exchange'): print('    exchange', {k: v for k, v in j['exchange'].items() if k.endswith('_ms')}, j['replicas_identical'])
PY
