#!/bin/bash
# the round's closing pass on one box: GPU suite, profile set of every config (+ the tile-200 step), bench lines
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_closing.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_closing.log
bash tools/r05_profiles.sh 51492c3 2>&1 | tail -3
bash tools/profile_round.sh r05_t200 51492c3 c3_300k_800 --tile 200 > gpurun_out/r05_t200.log 2>&1; mkdir -p gpurun_out/r05_t200 && cp gpurun_out/prof_r05_t200/r05_t200_*.json gpurun_out/prof_r05_t200/r05_t200_kernel_stats.csv gpurun_out/r05_t200/ && rm -rf gpurun_out/prof_r05_t200
# (the lines look for their counters in profiles/: the fresh summaries go there first)
cp gpurun_out/r05_profiles/r05_*_{kernel_stats.csv,hbm_traffic_pmc.json,sq_counters.json,lane_counters.json,blend_isa_mix.json} gpurun_out/r05_t200/r05_t200_* profiles/
bash tools/r05_lines.sh 2>&1 | tail -14
