#!/bin/bash
# kernel trace around a planned densify event (single-device step)
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r06_event; rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 tools/densify_host_timeline.py 1 > $out/timeline.txt 2>&1 || { tail -5 $out/timeline.txt; exit 1; }
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $out/event_kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
ev = [i for i, n in enumerate(names) if 'dn_' in n or 'densify' in n.lower()]
print('kernels', len(rows), 'densify kernels', len(ev))
# the last event (rep 2): take the last cluster of densify kernels
last = ev[-1]
first = last
while first - 1 in ev or (first - 1 >= 0 and any(e >= first - 12 and e < first for e in ev)):
    first = max(e for e in ev if e < first)
    if first == ev[0]: break
lo, hi = max(0, first - 25), min(len(rows), last + 40)
t0 = int(rows[lo]['Start_Timestamp'])
prev_end = None
for r in rows[lo:hi]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    gap = (s - prev_end) if prev_end is not None else 0
    print(f"{s/1000:9.1f} {e/1000:9.1f} dur {(e-s)/1000:7.1f} gap {gap/1000:7.1f}  {r['Kernel_Name'][:90]}")
    prev_end = max(prev_end or 0, e)
PY
tail -4 $out/timeline.txt
