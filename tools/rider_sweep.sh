#!/bin/bash
# shares of the colour riders per host kernel, one box.  GSPLAT_RIDER_SHARES = permille of a forward's colour units for
# the GS_RIDE_HOSTS = 3 host kernels in the enum's order (gs_ctx.h): ss_hist, ss_scatter, wide_tile; what is left runs as
# colour_rest_kernel.  (bucket_sort was a host once: measured and dropped, DESIGN section 4 "Colour riders".)
# usage: bash tools/rider_sweep.sh [config] ["a,b,c" ...]
cfg=${1:-c3_300k_800}; shift
sets=("$@"); [ ${#sets[@]} -eq 0 ] && sets=("350,450,200" "450,550,0" "300,400,300" "250,350,150" "400,400,200" "200,300,200" "350,450,0")
python tools/rider_ab.py $cfg 0,0 2>/dev/null | tail -n 1
for sh in "${sets[@]}"; do
  echo "shares $sh"
  GSPLAT_RIDER_SHARES=$sh python tools/rider_ab.py $cfg 1,1 2>/dev/null | tail -n 1
done
python tools/rider_ab.py $cfg 0 2>/dev/null | tail -n 1
