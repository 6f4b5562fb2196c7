for cap in 25165824 8388608 5242880; do
  GSPLAT_BENCH_PAIR_CAP=$cap python bench.py --no-cpu-baseline --steps 40 > gpurun_out/cap_$cap.json 2>/dev/null
  python - $cap <<'PY'
import json,sys
d=json.load(open(f"gpurun_out/cap_{sys.argv[1]}.json"))
print(sys.argv[1], "views/s", d["value"], {k: v["ms"] for k, v in d["stages"].items()}, d["config"].get("workload_stats", d.get("workload_stats")))
PY
done
