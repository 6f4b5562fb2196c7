#!/bin/bash
# Regenerates the rocprofv3 summaries kept under profiles/ (run on the GPU box through gpurun; writes under gpurun_out/).
# usage: tools/profile_round.sh <tag>     e.g. r01c
tag=${1:-r01}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 $root/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $out/stats_bench.json 2> $out/stats.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o p -- python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/pmc_$c.json 2> $out/pmc_$c.log
done
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/pmc_SQ -o p -- python3 $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $out/pmc_SQ.json 2> $out/pmc_SQ.log
python3 - "$out" "$tag" <<'PY'
import sys, glob, csv, json, collections, shutil, os
out, tag = sys.argv[1], sys.argv[2]
st = glob.glob(out + "/stats/*kernel_stats.csv")
if st:
    shutil.copy(st[0], f"{out}/{tag}_kernel_stats.csv")
def collect(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            a = acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    return acc
kern = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, d in collect(f"{out}/pmc_{c}").items():
        if k.startswith("gs::") or "gs::" in k:
            v = d[c]
            kern[k][f"{c}_KB_per_launch"] = round(v[0] / max(v[1], 1), 1)
            kern[k][f"launches_{c}"] = v[1]
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, two separate passes of `bench.py --steps 6 --warmup 2` on MI355X. "
                   "Per the MI355X guide FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads on gfx950: double it before "
                   "comparing with byte counts; WRITE_SIZE is exact for 16-B stores and float atomics.", "kernels": kern},
          open(f"{out}/{tag}_hbm_traffic_pmc.json", "w"), indent=1)
sq = {k: {c: round(v[0] / max(v[1], 1), 1) for c, v in d.items()} for k, d in collect(f"{out}/pmc_SQ").items() if "gs::" in k}
json.dump({"note": "rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU, per-launch averages. "
                   "VALU busy = 4 * SQ_ACTIVE_INST_VALU / 1024 SIMDs / (SQ_BUSY_CYCLES / 32 shader engines).", "kernels": sq},
          open(f"{out}/{tag}_sq_counters.json", "w"), indent=1)
print("written", os.listdir(out))
PY
