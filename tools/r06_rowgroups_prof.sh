#!/bin/bash
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/r06_rowgroups_prof; rm -rf $out; mkdir -p $out
for t in 1 2; do
  export GSPLAT_TRIM_RECTS=$t
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/t$t -o s -- python3 $root/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_t$t.json 2> $out/err_t$t.txt || { tail -3 $out/err_t$t.txt; exit 1; }
  f=$(find $out/t$t -name "*kernel_stats.csv" | head -1); cp $f $out/stats_t$t.csv; rm -rf $out/t$t
done
python3 - <<'PY'
import csv, os
root=os.environ['GRAFT_REPO_ROOT']
d={}
for t in (1,2):
    for r in csv.DictReader(open(f'{root}/gpurun_out/r06_rowgroups_prof/stats_t{t}.csv')):
        d.setdefault(r['Name'].split('(')[0].replace('void gs::','').replace('gs::','')[:48],{})[t]=(float(r['AverageNs'])/1000,int(r['Calls']))
for k,v in sorted(d.items(), key=lambda kv:-kv[1].get(1,(0,0))[0]*kv[1].get(1,(0,0))[1])[:24]:
    print(f"{k:50s}", v.get(1), v.get(2))
PY
