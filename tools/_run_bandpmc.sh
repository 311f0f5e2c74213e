set -e
mkdir -p gpurun_out/r03q
for wl in 50m 10m; do
  bash tools/pmc_cmd.sh band_${wl}_fetch "FETCH_SIZE" tools/band_bench.py --workload $wl --ranks 8 --steps 2 > gpurun_out/r03q/band_${wl}_fetch.txt
  bash tools/pmc_cmd.sh band_${wl}_write "WRITE_SIZE" tools/band_bench.py --workload $wl --ranks 8 --steps 2 > gpurun_out/r03q/band_${wl}_write.txt
done
grep -h preprocess gpurun_out/r03q/*.txt
