#!/bin/bash
# round 5: the planned densify event: tests, the event's cost (20-step driver line and 100 steps), with and without
out=gpurun_out/r05_densify; rm -rf $out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trajectory.py -m gpu -x -q -k "densify or split_and_prune or train_steps or reload or trajectory" > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $out/pytest.log
[ $rc -ne 0 ] && exit 1
run() { name=$1; shift; timeout -k 10 300 python bench.py "$@" --no-cpu-baseline > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; }; }
for rep in 1 2 3; do
  for d in 1 0; do
    GSPLAT_PLANNED_DENSIFY=$d run c3_20_p${d}_$rep --steps 20 --warmup 5
  done
done
for d in 1 0; do GSPLAT_PLANNED_DENSIFY=$d run c3_100_p$d --steps 100 --warmup 10; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05_densify/*.json')):
    try: j = json.load(open(f))
    except Exception as e: print(f, 'BAD', e); continue
    print(f.split('/')[-1], j['value'], j['unit'], j['ms_per_step'], j['densify']['last_stats'], j['densify']['N_after'])
PY
