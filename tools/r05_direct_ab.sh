#!/bin/bash
# round 5: direct tile scatter against expansion + one-pass tile sort: the binning tests, then bench lines alternating on one box
out=gpurun_out/r05_direct; rm -rf $out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_binning_large.py tests/test_gpu_parity.py -m gpu -x -q -k "tile_bin or direct or cut_binning or tile_sorts or fused_render_forward_backward or scatter or two_word or randomized" > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $out/pytest.log
[ $rc -ne 0 ] && exit 1
run() { name=$1; shift; timeout -k 10 300 python bench.py "$@" --no-cpu-baseline > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; }; }
for rep in 1 2; do
  for d in 1 0; do
    GSPLAT_DIRECT_SCATTER=$d run c3_d${d}_$rep --steps 100 --warmup 10
  done
done
for d in 1 0; do
  GSPLAT_DIRECT_SCATTER=$d run c1_d$d --config c1_10k_400 --steps 200 --warmup 20
  GSPLAT_DIRECT_SCATTER=$d run c2_d$d --config c2_100k_800 --steps 100 --warmup 10
  GSPLAT_DIRECT_SCATTER=$d run grown_d$d --config c3_grown_1m --steps 60 --warmup 10
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05_direct/*.json')):
    try: j = json.load(open(f))
    except Exception as e: print(f, 'BAD', e); continue
    print(f.split('/')[-1], j['value'], j['unit'], j['ms_per_step'], {k: v['ms'] for k, v in j['stages'].items()})
PY
