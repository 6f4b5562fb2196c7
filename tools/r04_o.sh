#!/bin/bash
for q in 8 1 8 1 4 2; do GSPLAT_BWD_QUEUES=$q python tools/bwd_ab.py c3_300k_800 2>/dev/null | tail -1; done
for q in 8 1; do GSPLAT_BWD_QUEUES=$q python tools/bwd_ab.py c2_100k_800 2>/dev/null | tail -1; done
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_o.log 2>&1; tail -3 gpurun_out/pytest_o.log
