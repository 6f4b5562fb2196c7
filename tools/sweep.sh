#!/bin/bash
# usage: tools/sweep.sh "<residency list>"  -- runs bench.py per setting and prints value / fwd / stage times
for r in $1; do
  python bench.py --no-cpu-baseline --steps 40 --residency $r > gpurun_out/sw_$r.json 2>/dev/null
  python - "$r" <<'PY'
import json,sys
r=sys.argv[1]
d=json.load(open(f"gpurun_out/sw_{r}.json"))
print(r, "views/s", d["value"], "fwd_ms", d["fwd_ms"], {k: v["ms"] for k, v in d["stages"].items()}, d.get("depth_cuts"), d["workload_stats"]["M_pairs"])
PY
done
