#!/bin/bash
# development aid: the UNet's GroupNorm shapes at batch B under the shuffle-reduced kernel (default) and without it (SR_GN_WAVE_MAX_WG=0)
for B in "$@"; do
  for shape in "64 1280" "64 2560" "256 1280" "256 2560" "256 1920" "1024 640" "1024 1280" "1024 1920" "1024 960" "4096 320" "4096 640" "4096 960"; do
    set -- $shape
    a=$(python tools/bench_gn.py $B $1 $2 | awk '{print $4}')
    b=$(SR_GN_WAVE_MAX_WG=0 python tools/bench_gn.py $B $1 $2 | awk '{print $4}')
    echo "B$B HW$1 C$2: wave $a us, before $b us"
  done
done
