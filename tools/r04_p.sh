#!/bin/bash
for q in 8 1 8 1 4; do GSPLAT_FWD_QUEUES=$q python tools/bwd_ab.py c3_300k_800 2>/dev/null | tail -1; done
for q in 8 1; do GSPLAT_FWD_QUEUES=$q python tools/bwd_ab.py c2_100k_800 2>/dev/null | tail -1; done
for q in 8 1; do GSPLAT_FWD_QUEUES=$q python tools/bwd_ab.py c1_10k_400 2>/dev/null | tail -1; done
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_p.log 2>&1; tail -3 gpurun_out/pytest_p.log
bash tools/pmc_fwd.sh FETCH_SIZE 2>&1 | grep -E "blend"
