#!/bin/bash
# per-kernel time of an arbitrary python command: tools/kstats_cmd.sh <tag> <script> [args]   (GPU box, through gpurun)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/kstats_$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/s -o s -- python3 $GRAFT_REPO_ROOT/"$@" > $out/out.txt 2> $out/log.txt || { tail -5 $out/log.txt; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/s/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:30]:
    print(f"{r['Name'].split('(')[0][:58]:58s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.2f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms {float(r['Percentage']):5.2f}%")
PY
