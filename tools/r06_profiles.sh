#!/bin/bash
# round-6 profile set: kernel stats + HBM traffic + SQ + lane counters per config (summaries only travel back), the binning critical
# path of c3, kernel resources, the 1-rank native DP step's kernel stats
commit=$1
keep=gpurun_out/r06_profiles; rm -rf $keep; mkdir -p $keep
run() {   # tag config extra...
  tag=$1; cfg=$2; shift 2
  bash tools/profile_round.sh $tag $commit $cfg "$@" > $keep/$tag.log 2>&1 || { echo "$tag FAILED"; tail -5 $keep/$tag.log; return 1; }
  cp gpurun_out/prof_$tag/${tag}_*.json gpurun_out/prof_$tag/${tag}_kernel_stats.csv $keep/ 2>/dev/null
  cp gpurun_out/prof_$tag/stats_bench.json $keep/${tag}_bench_line_of_the_stats_run.json 2>/dev/null
  rm -rf gpurun_out/prof_$tag
  echo "$tag done"
}
run r06_c3 c3_300k_800 && run r06_grown c3_grown_1m && run r06_c5 c5_garden_2m --views 8 && run r06_c2 c2_100k_800 && run r06_c1 c1_10k_400 && run r06_t200 c3_300k_800 --tile 200
python tools/kernel_resources.py --out $keep/r06_kernel_resources.json > /dev/null 2>&1
bash tools/r06_binpath.sh > $keep/binpath.log 2>&1; cp gpurun_out/r06_binpath/r06_binning_critical_path.json $keep/r06_binning_critical_path_c3.json; rm -rf gpurun_out/r06_binpath
root=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$keep/stats_dp1 -o s -- python3 $root/bench.py --steps 30 --warmup 5 --dp-single --dp-impl native --no-cpu-baseline > $root/$keep/r06_dp1_native_bench_line_of_the_stats_run.json 2> $root/$keep/stats_dp1.log )
cp $(find $keep/stats_dp1 -name "*kernel_stats.csv" | head -1) $keep/r06_dp1_native_kernel_stats.csv; rm -rf $keep/stats_dp1
ls $keep
