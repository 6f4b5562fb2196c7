#!/bin/bash
for b in 1 0; do
GSPLAT_COLOUR_RIDERS=3 GSPLAT_SS_BIG=$b bash tools/kstats_cmd.sh c5fwd_b$b tools/fwd_only.py c5_garden_2m 40 4 2>&1 | grep -E "ss_|bucket|radix|colscan|proj_fwd|expand|scan_block|prefix|wide_|chunk" 
echo ----
done
