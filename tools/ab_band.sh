#!/bin/bash
# same-box A/B of a rank's band between this tree's library and another build (GS3D_LIB):
#   tools/ab_band.sh build/ab/libgs3d_hip_<rev>.so "10m 50m" 8
old=$1; wls=${2:-"10m 50m"}; ranks=${3:-8}
for wl in $wls; do
  for rep in 1 2; do
    echo "== $wl ranks $ranks new"; python3 tools/band_bench.py --workload $wl --ranks $ranks || exit 1
    echo "== $wl ranks $ranks old"; GS3D_LIB=$old python3 tools/band_bench.py --workload $wl --ranks $ranks || exit 1
  done
done
