#!/bin/bash
# the bench lines of every single-GPU config, one box: tools/bench_all.sh <tag>   (GPU box, through gpurun; writes gpurun_out/<tag>_bench_*.json)
tag=${1:-rXX}; o=$GRAFT_REPO_ROOT/gpurun_out
python bench.py --steps 100 --warmup 10 2>/dev/null | tail -n 1 > $o/${tag}_bench_default.json || exit 1
python bench.py --config c1_10k_400 2>/dev/null | tail -n 1 > $o/${tag}_bench_c1_10k_400_forward.json || exit 1
python bench.py --config c2_100k_800 2>/dev/null | tail -n 1 > $o/${tag}_bench_c2_100k_800_fwdbwd.json || exit 1
python bench.py --config c5_garden_2m --steps 240 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -n 1 > $o/${tag}_bench_c5_garden_2m_240steps.json || exit 1
python bench.py --tile 200 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -n 1 > $o/${tag}_bench_tile200.json || exit 1
python - $o $tag <<'PY'
import json, sys, glob
for f in sorted(glob.glob(f"{sys.argv[1]}/{sys.argv[2]}_bench_*.json")):
    d = json.loads(open(f).read())
    print(f.split("/")[-1], d["value"], d["unit"], d["ms_per_step"], {k: v["ms"] for k, v in d.get("stages", {}).items()})
PY
