#!/bin/bash
# shares of the colour riders per host kernel (GSPLAT_RIDER_SHARES: ss_hist, ss_scatter, bucket_sort, wide_tile, lsd_hist, lsd_scatter, tiny), one box
cfg=${1:-c3_300k_800}
python tools/rider_ab.py $cfg 0,0 2>/dev/null | tail -n 1
for sh in "250,300,300,150" "300,350,200,150" "200,250,400,150" "350,450,0,200" "250,350,250,150" "200,300,300,200" "300,400,300,0"; do
  echo "shares $sh"
  GSPLAT_RIDER_SHARES=$sh python tools/rider_ab.py $cfg 1,1 2>/dev/null | tail -n 1
done
python tools/rider_ab.py $cfg 0 2>/dev/null | tail -n 1
