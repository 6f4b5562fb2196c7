#!/bin/bash
# round 6: robustness runs on the final kernels -- adversarial fuzz (16x16 and block lists; one-wave / pair forwards too), the soak
out=gpurun_out/r06_robust; rm -rf $out; mkdir -p $out
timeout -k 10 600 python tools/fuzz_parity.py 1000 140000 > $out/r06_fuzz_1000_seeds_140000.txt 2>&1; echo "fuzz rc=$?"; tail -3 $out/r06_fuzz_1000_seeds_140000.txt
FUZZ_TILES=24,40,100,200,7,33 timeout -k 10 400 python tools/fuzz_parity.py 500 145000 > $out/r06_fuzz_block_lists_500_seeds_145000.txt 2>&1; echo "fuzz bl rc=$?"; tail -3 $out/r06_fuzz_block_lists_500_seeds_145000.txt
timeout -k 10 600 python tools/soak.py 5000 > $out/r06_soak_5000_steps_final.txt 2>&1; echo "soak rc=$?"; tail -3 $out/r06_soak_5000_steps_final.txt
