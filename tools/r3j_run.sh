mkdir -p gpurun_out/r3j; rm -f gpurun_out/r3j/*
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reload or profiler or train_steps" > gpurun_out/r3j/t.log 2>&1; echo "rc=$?" >> gpurun_out/r3j/t.log
tail -n 6 gpurun_out/r3j/t.log
