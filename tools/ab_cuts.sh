#!/bin/bash
# A/B of the depth cuts on the bench workload: alternating runs in one box
for i in 1 2 3; do
  for flag in "" "--no-depth-cuts"; do
    python bench.py --no-cpu-baseline --steps 80 --warmup 16 $flag > gpurun_out/ab.json 2>/dev/null
    python - "$flag" <<'PY'
import json,sys
d=json.load(open("gpurun_out/ab.json"))
print((sys.argv[1] or "cuts").ljust(16), "views/s", d["value"], "ms/step", d["ms_per_step"], d["step_ms_spread"], "bin", d["stages"]["bin"]["ms"], d["depth_cuts"])
PY
  done
done
