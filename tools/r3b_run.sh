mkdir -p gpurun_out/r3b
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "native or fused_render or op_level or blend_forward or trainer_exchanges or overflow or randomized" --durations=5 > gpurun_out/r3b/new_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3b/new_tests.log
for m in forward fwdbwd train; do python bench.py --tile 200 --mode $m --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3b/bench_tile200_$m.json 2> gpurun_out/r3b/bench_tile200_$m.err; done
tail -5 gpurun_out/r3b/new_tests.log
