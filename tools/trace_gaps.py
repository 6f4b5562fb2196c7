"""Idle time between kernels in a rocprofv3 kernel trace: python tools/trace_gaps.py <kernel_trace.csv> [last_n_kernels]
Merges all streams; a gap is time in which NO kernel runs.  Prints the gaps summed by (kernel before -> kernel after)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gs::", "")[:40]) for r in rows]
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
rows = rows[-n:]
gaps = collections.defaultdict(lambda: [0.0, 0])
busy_end = rows[0][1]; prev = rows[0][2]; idle = 0.0
for s, e, name in rows[1:]:
    if s > busy_end:
        g = (s - busy_end) / 1e3
        gaps[(prev, name)][0] += g; gaps[(prev, name)][1] += 1; idle += g
    if e > busy_end:
        busy_end = e; prev = name
span = (rows[-1][1] - rows[0][0]) / 1e3
print(f"span {span / 1e3:.2f} ms, idle {idle / 1e3:.2f} ms = {idle / span:.1%}; kernels {len(rows)}")
for (a, b), (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"{t / 1e3:8.3f} ms in {c:5d} gaps (avg {t / c:6.1f} us)  {a} -> {b}")
