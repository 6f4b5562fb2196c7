mkdir -p gpurun_out/r3c
python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r3c/gpu_all.log 2>&1; echo "rc=$?" >> gpurun_out/r3c/gpu_all.log
python bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r3c/bench_default.json 2> gpurun_out/r3c/bench_default.err
python bench.py --config c5_garden_2m --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/r3c/bench_c5.json 2> gpurun_out/r3c/bench_c5.err
tail -4 gpurun_out/r3c/gpu_all.log
