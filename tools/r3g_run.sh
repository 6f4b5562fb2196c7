mkdir -p gpurun_out/r3g
timeout -k 10 400 python tools/dp_overflow_rehearsal.py > gpurun_out/r3g/dp_overflow.log 2>&1; echo "rc=$?" >> gpurun_out/r3g/dp_overflow.log
GSPLAT_BENCH_DEVICE=0 timeout -k 10 400 python bench.py --gpus 2 --backend gloo --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r3g/bench_dp2_gloo.json 2> gpurun_out/r3g/bench_dp2_gloo.err; echo "rc=$?" >> gpurun_out/r3g/bench_dp2_gloo.err
grep -a "REHEARSAL\|exit codes\|rc=\|Error\|error" gpurun_out/r3g/dp_overflow.log | tail -n 8
tail -n 3 gpurun_out/r3g/bench_dp2_gloo.err; head -c 400 gpurun_out/r3g/bench_dp2_gloo.json
