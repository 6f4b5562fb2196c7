#!/bin/bash
out=gpurun_out/r04i; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err && echo c3 ok
python bench.py --config c2_100k_800 --steps 100 --warmup 10 --no-cpu-baseline > $out/bench_c2.json 2> $out/bench_c2.err && echo c2 ok
python bench.py --config c1_10k_400 --steps 100 --warmup 10 --no-cpu-baseline > $out/bench_c1.json 2> $out/bench_c1.err && echo c1 ok
python bench.py --config c3_grown_1m --steps 90 --warmup 10 --no-cpu-baseline > $out/bench_grown.json 2> $out/bench_grown.err && echo grown ok
bash tools/pmc_fwd.sh FETCH_SIZE WRITE_SIZE > $out/pmc_fwd.txt 2>&1; cat $out/pmc_fwd.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04i/bench_*.json')):
    j = json.load(open(f))
    print(f.split('/')[-1], j['value'], j['ms_per_step'], 'fwd_ms', j['fwd_ms'], {k: v['ms'] for k, v in j['stages'].items()})
PY
