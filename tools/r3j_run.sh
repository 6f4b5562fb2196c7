mkdir -p gpurun_out/r3j; rm -f gpurun_out/r3j/*
timeout -k 10 600 python -m pytest tests/test_gpu_binning_large.py -x -q -m gpu > gpurun_out/r3j/bin_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3j/bin_tests.log
tail -n 4 gpurun_out/r3j/bin_tests.log
