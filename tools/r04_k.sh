#!/bin/bash
out=gpurun_out/r04k; mkdir -p $out
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --cut-min-dropped 1000000 > $out/bench_c3_cuts.json 2> $out/err.txt && echo cuts ok
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $out/bench_c3.json 2> $out/err.txt && echo nocuts ok
bash tools/kstats_cmd.sh c3cuts bench.py --steps 60 --warmup 10 --no-cpu-baseline --cut-min-dropped 1000000 > $out/kstats_cuts.txt 2>&1; head -32 $out/kstats_cuts.txt
python - <<'PY'
import json
for f in ('bench_c3_cuts', 'bench_c3'):
    j = json.load(open('gpurun_out/r04k/' + f + '.json'))
    print(f, j['value'], j['ms_per_step'], {k: v['ms'] for k, v in j['stages'].items()}, j['workload_stats']['M_pairs'], j['depth_cuts'], j['step_ms_spread'])
PY
