mkdir -p gpurun_out/r3h
python -m pytest tests/test_gpu_binning_large.py tests/test_gpu_parity.py -x -q -m gpu -k "bin or binning or two_word or cut or reserve or randomized or garden or full_size" > gpurun_out/r3h/tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3h/tests.log
for i in 1 2; do python bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r3h/b_$i.json 2>/dev/null; done
python bench.py --config c5_garden_2m --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r3h/b_c5.json 2>/dev/null
tail -n 3 gpurun_out/r3h/tests.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3h/b_*.json')):
    j=json.load(open(f)); print(f, j['value'], {k:v['ms'] for k,v in j['stages'].items()})
PY
