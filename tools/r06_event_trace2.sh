#!/bin/bash
# kernel trace of the driver's command (20 steps, 5 warmup): the densify event inside it
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r06_event2; rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.txt 2>&1 || { tail -5 $out/bench.txt; exit 1; }
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $out/event_kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
cl = [i for i, n in enumerate(names) if 'classify_kernel' in n]
print('kernels', len(rows), 'classify at', cl)
for c in cl[-1:]:
    lo, hi = max(0, c - 30), min(len(rows), c + 60)
    t0 = int(rows[lo]['Start_Timestamp'])
    prev_end = None
    for r in rows[lo:hi]:
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        gap = (s - prev_end) if prev_end is not None else 0
        print(f"{s/1000:9.1f} {e/1000:9.1f} dur {(e-s)/1000:7.1f} gap {gap/1000:7.1f}  {r['Kernel_Name'][:80]}")
        prev_end = max(prev_end or 0, e)
# gaps over the whole timed region's tail: sum of gaps > 3 us in the last 25 steps
PY
grep -v "^[EWI]2026" $out/bench.txt | tail -1 | cut -c1-300
