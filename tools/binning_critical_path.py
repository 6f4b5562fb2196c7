#!/usr/bin/env python3
"""The binning stage's critical path, per launch (round 6, the verdict's item 7): duration, occupancy of the launch (workgroups
against the 256 CUs; waves against the chip's 1024 SIMDs), the gap to the NEXT launch, and the HBM-side bytes of the PMC passes.

    python tools/binning_critical_path.py <kernel_trace.csv> [--fetch <counter_collection.csv>] [--write <counter_collection.csv>]
                                          [--out profiles/r06_binning_critical_path.json]

kernel_trace.csv: rocprofv3 --kernel-trace of `bench.py --steps 30 --warmup 5` (Start_Timestamp / End_Timestamp / Grid_Size /
Workgroup_Size per dispatch).  The binning launches of a step are the dispatches between the projection
(proj_fwd_fused_kernel) and the blend forward (blend_fwd_v2*); steps whose chain differs from the most common one (first
visits, cut forwards) are left out; the numbers are medians over the steps that remain."""
import argparse, collections, csv, json, statistics, sys

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--fetch"); ap.add_argument("--write"); ap.add_argument("--out")
ap.add_argument("--meta", default="{}")
a = ap.parse_args()


def short(n):
    return n.split("(")[0].replace("void ", "").replace("gs::", "")


rows = []
for r in csv.DictReader(open(a.trace)):
    g = [int(r.get(k, 1) or 1) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")] if "Grid_Size_X" in r else [int(r["Grid_Size"]), 1, 1]
    w = [int(r.get(k, 1) or 1) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z")] if "Workgroup_Size_X" in r else [int(r["Workgroup_Size"]), 1, 1]
    rows.append(dict(s=int(r["Start_Timestamp"]), e=int(r["End_Timestamp"]), name=short(r["Kernel_Name"]), grid=g[0] * g[1] * g[2], wg=w[0] * w[1] * w[2],
                     lds=int(r.get("LDS_Block_Size", 0) or 0), vgpr=int(r.get("VGPR_Count", 0) or 0)))
rows.sort(key=lambda r: r["s"])
# steps: from a projection kernel to the next blend forward
steps, cur = [], None
for r in rows:
    if r["name"].startswith("proj_fwd_fused_kernel"):
        cur = [r]
    elif cur is not None:
        cur.append(r)
        if r["name"].startswith("blend_fwd_v2"):
            steps.append(cur); cur = None
chains = collections.Counter(tuple(x["name"] for x in s) for s in steps)
chain, n = chains.most_common(1)[0]
sel = [s for s in steps if tuple(x["name"] for x in s) == chain]
pmc = {}
for key, path in (("fetch", a.fetch), ("write", a.write)):
    if not path:
        continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        v = acc[short(r["Kernel_Name"])]
        v[0] += float(r["Counter_Value"]); v[1] += 1
    pmc[key] = {k: v[0] / max(v[1], 1) for k, v in acc.items()}
out = []
for i, name in enumerate(chain):
    d = [s[i]["e"] - s[i]["s"] for s in sel]
    gap = [s[i + 1]["s"] - s[i]["e"] for s in sel] if i + 1 < len(chain) else [0]
    r0 = sel[0][i]
    wgs = r0["grid"] // max(r0["wg"], 1)
    waves = wgs * ((r0["wg"] + 63) // 64)
    ent = dict(launch=i, kernel=name, workgroups=wgs, threads_per_workgroup=r0["wg"], waves=waves,
               workgroups_per_cu=round(wgs / 256.0, 2), waves_per_simd=round(waves / 1024.0, 2),
               duration_us=round(statistics.median(d) / 1e3, 2), gap_to_next_us=round(statistics.median(gap) / 1e3, 2))
    if "fetch" in pmc and name in pmc["fetch"]:
        ent["fetch_MB"] = round(2.0 * pmc["fetch"][name] * 1024 / 1e6, 2)      # FETCH_SIZE is KB and reads half on gfx950 (MI355X guide)
    if "write" in pmc and name in pmc["write"]:
        ent["write_MB"] = round(pmc["write"][name] * 1024 / 1e6, 2)
    if "fetch_MB" in ent and "write_MB" in ent:
        ent["GBps"] = round((ent["fetch_MB"] + ent["write_MB"]) / ent["duration_us"] * 1e3 / 1e3, 1)
    out.append(ent)
first, last = 1, len(chain) - 2          # the binning launches: between the projection and the blend forward
span = statistics.median([s[last + 1]["s"] - s[first]["s"] for s in sel]) / 1e3 if last >= first else 0.0
res = dict(json.loads(a.meta), note="one row per launch of the step's forward up to the blend (median over the steps with the most common chain); "
           "the binning stage = the launches between proj_fwd_fused_kernel and blend_fwd_v2*; fetch_MB = 2 x FETCH_SIZE (gfx950), "
           "write_MB = WRITE_SIZE; PMC averages are per kernel NAME over the counter pass (a name launched with several shapes is averaged)",
           steps_used=len(sel), steps_seen=len(steps), binning_span_us=round(span, 2),
           binning_kernel_sum_us=round(sum(e["duration_us"] for e in out[first:last + 1]), 2),
           binning_gap_sum_us=round(sum(e["gap_to_next_us"] for e in out[first - 1:last + 1]), 2), launches=out)
txt = json.dumps(res, indent=1)
if a.out:
    open(a.out, "w").write(txt)
for e in out:
    print(f"{e['launch']:2d} {e['kernel'][:44]:44s} wgs {e['workgroups']:6d} x {e['threads_per_workgroup']:4d}  {e['workgroups_per_cu']:6.2f} wg/CU {e['waves_per_simd']:6.2f} waves/SIMD  "
          f"{e['duration_us']:7.2f} us  gap {e['gap_to_next_us']:5.2f}  fetch {e.get('fetch_MB', '-')} write {e.get('write_MB', '-')} MB")
print("binning span", res["binning_span_us"], "us; kernel sum", res["binning_kernel_sum_us"], "gaps", res["binning_gap_sum_us"], "steps", len(sel), "/", len(steps))
