#!/bin/bash
# one rank's band (the middle band of an equal-rows plan) on ONE GPU, for 2 / 4 / 8 ranks and every workload:
#   tools/band_table.sh > gpurun_out/band_costs.log
for wl in 1m 10m 10m-4k 50m; do
  for ranks in 2 4 8; do
    python3 tools/band_bench.py --workload $wl --ranks $ranks 2>/dev/null | grep "^{" || exit 1
  done
done
