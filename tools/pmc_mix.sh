#!/bin/bash
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/pmc_mix; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --no-cpu-baseline --steps 6 --warmup 2"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 --kernel-trace --output-format csv -d $out/a -o p -- $B > $out/a.json 2> $out/a.log || { tail -5 $out/a.log; exit 1; }
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/b -o p -- $B > $out/b.json 2> $out/b.log || { tail -5 $out/b.log; exit 1; }
python3 - "$out" <<'PY'
import sys, glob, csv, json, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "blend_" in k:
            a = acc[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print(k[:50]); print("   ", {c: round(v[0] / max(v[1], 1), 1) for c, v in d.items()})
PY
