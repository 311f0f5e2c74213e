set -e
export TMPDIR=/tmp
ROOT=$PWD
mkdir -p gpurun_out/r03q
cd /tmp
for v in 1 0; do
  GS3D_BLOCK_LIST=$v rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03q/prof_bl$v -o p -- python3 $ROOT/bench.py --workload 50m --no-roofline --no-cpu-baseline --no-in-flight --extra-workloads "" --steps 10 --warmup 3 --frame-samples 0 > $ROOT/gpurun_out/r03q/prof_bl$v.log 2>&1
  f=$(find $ROOT/gpurun_out/r03q/prof_bl$v -name "*kernel_stats.csv" | head -1)
  echo "BLOCK_LIST=$v"; grep -E "k_preprocess|k_block_cull" "${f:-/dev/null}" | cut -c1-200
done
find $ROOT/gpurun_out/r03q -name "*.csv" -size +1M -delete
