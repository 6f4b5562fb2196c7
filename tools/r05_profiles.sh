#!/bin/bash
# round-4 profile set: kernel stats + HBM traffic + SQ + lane counters for the headline config and the others.
# Only the summaries travel back (gpurun merges at most 64 MiB; the raw traces are far larger).
commit=$1
keep=gpurun_out/r05_profiles; mkdir -p $keep
run() {   # tag config extra...
  tag=$1; cfg=$2; shift 2
  bash tools/profile_round.sh $tag $commit $cfg "$@" > $keep/$tag.log 2>&1 || { echo "$tag FAILED"; tail -5 $keep/$tag.log; return 1; }
  cp gpurun_out/prof_$tag/${tag}_*.json gpurun_out/prof_$tag/${tag}_kernel_stats.csv $keep/ 2>/dev/null
  cp gpurun_out/prof_$tag/stats_bench.json $keep/${tag}_bench_line_of_the_stats_run.json 2>/dev/null
  rm -rf gpurun_out/prof_$tag
  echo "$tag done"
}
run r05_c3 c3_300k_800 && run r05_grown c3_grown_1m && run r05_c5 c5_garden_2m --views 8 && run r05_c2 c2_100k_800 && run r05_c1 c1_10k_400
ls $keep
