#!/bin/bash
for q in 8 4 2 1 8 1; do GSPLAT_FWD_QUEUES=$q python tools/bwd_ab.py c3_300k_800 2>/dev/null | tail -1; done
for q in 8 1; do GSPLAT_FWD_QUEUES=$q python tools/bwd_ab.py c2_100k_800 2>/dev/null | tail -1; done
