set -e
mkdir -p gpurun_out/r03q
for wl in 50m 10m 10m-4k; do
 for v in 0 1; do
  echo "BLOCK_LIST=$v $wl" | tee -a gpurun_out/r03q/blocklist.log
  GS3D_BLOCK_LIST=$v python tools/band_bench.py --workload $wl --ranks 8 2>&1 | tail -1 | tee -a gpurun_out/r03q/blocklist.log
  GS3D_BLOCK_LIST=$v python tools/band_bench.py --workload $wl --ranks 2 2>&1 | tail -1 | tee -a gpurun_out/r03q/blocklist.log
 done
done
python tools/ab_bench.py --workloads 10m,50m --set list: --set inkernel:GS3D_BLOCK_LIST=0 --repeat 2 | tee -a gpurun_out/r03q/blocklist.log
python -m pytest tests/test_gpu_render.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r03q/tests_blocklist.log
