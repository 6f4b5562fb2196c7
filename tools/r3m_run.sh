mkdir -p gpurun_out/r3m; rm -f gpurun_out/r3m/*
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3m/gpu_all.log 2>&1; echo "rc=$?" >> gpurun_out/r3m/gpu_all.log
tail -n 3 gpurun_out/r3m/gpu_all.log
for c in c1_10k_400 c2_100k_800 c3_300k_800 c5_garden_2m; do python bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r3m/$c.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3m/*.json')):
    try:
        j=json.load(open(f)); print(f, j['value'], j['unit'], j['ms_per_step'], {k:v['ms'] for k,v in j['stages'].items()}, j['fwd_mpix_per_s'])
    except Exception as e: print(f,'ERR',e)
PY
