#!/bin/bash
# the trimmed rects cut into four row groups (GS_TUNE_TRIM_RECTS = 2) against the box alone (1): tests, then A/B on one box
out=gpurun_out/r06_rowgroups; rm -rf $out; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "trimmed or randomized_small or adversarial or full_size_properties or garden or bench_workload or config1 or config2 or fused_render" > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
grep -q " failed" $out/pytest.txt && exit 1
for rep in 1 2; do for t in 2 1; do
  GSPLAT_TRIM_RECTS=$t timeout -k 10 200 python bench.py --steps 100 --warmup 30 --no-cpu-baseline > $out/c3_trim${t}_$rep.json 2>$out/err.txt || exit 1
done; done
for t in 2 1; do
  GSPLAT_TRIM_RECTS=$t timeout -k 10 300 python bench.py --config c3_grown_1m --steps 100 --no-cpu-baseline > $out/grown_trim${t}.json 2>>$out/err.txt || exit 1
  GSPLAT_TRIM_RECTS=$t timeout -k 10 300 python bench.py --config c5_garden_2m --steps 60 --no-cpu-baseline > $out/c5_trim${t}.json 2>>$out/err.txt || exit 1
  GSPLAT_TRIM_RECTS=$t timeout -k 10 300 python bench.py --config c2_100k_800 --steps 100 --no-cpu-baseline > $out/c2_trim${t}.json 2>>$out/err.txt || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_rowgroups/*.json')):
    j = json.load(open(f))
    print(f.split('/')[-1], j['value'], j['ms_per_step'], {k: v['ms'] for k, v in j['stages'].items()}, (j.get('workload_stats') or {}).get('M_pairs'))
PY
