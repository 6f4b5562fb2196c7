#!/bin/bash
# kernel-trace stats of one bench configuration, summary only:  bash tools/kstats.sh <tag> <bench args...>
tag=$1; shift
out=gpurun_out/ks_$tag; rm -rf $out; mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 bench.py "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$out/bench.json" <<'P'
import csv, sys, json
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Name"].startswith(("void gs", "gs::"))]
for r in rows:
    if int(r["Calls"]) >= 20: print(r["Name"][:64].ljust(64), r["Calls"].rjust(6), f'{float(r["AverageNs"]) / 1000:8.1f}')
try:
    d = json.load(open(sys.argv[2])); print(d["value"], d["unit"], d["ms_per_step"])
except Exception as e: print("bench line:", e)
P
cp "$f" gpurun_out/ks_$tag.csv; rm -rf $out
