#!/bin/bash
# PMC passes over a short bench run; prints per-kernel averages for the blend kernels.  usage: tools/pmc_fwd.sh "<counters pass 1>" "<counters pass 2>" ...
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "$@"; do
  i=$((i+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$i
  rm -rf $out
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$i.log 2>&1
  python3 - "$out" <<'PY'
import sys, glob, csv, collections
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "blend" not in k:
            continue
        a = acc[k.split("(")[0][-40:]][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print(k, {c: round(v[0] / max(v[1], 1), 1) for c, v in d.items()}, "launches", max(v[1] for v in d.values()))
PY
done
