mkdir -p gpurun_out/r3i; rm -f gpurun_out/r3i/*
python -m pytest tests -x -q -m gpu > gpurun_out/r3i/gpu_all.log 2>&1; echo "rc=$?" >> gpurun_out/r3i/gpu_all.log
for i in 1 2; do
  python bench.py --config c1_10k_400 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r3i/c1_$i.json 2>/dev/null
  python bench.py --config c2_100k_800 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r3i/c2_$i.json 2>/dev/null
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r3i/c3_$i.json 2>/dev/null
done
python bench.py --config c5_garden_2m --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r3i/c5_1.json 2>/dev/null
tail -n 3 gpurun_out/r3i/gpu_all.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3i/*.json')):
    try:
        j=json.load(open(f)); print(f, j['value'], j['unit'], {k:v['ms'] for k,v in j['stages'].items()}, j['workspace']['bytes']>>20)
    except Exception as e: print(f,'ERR',e)
PY
