set -e
export TMPDIR=/tmp
ROOT=$PWD
mkdir -p gpurun_out/r03q
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03q/trace_band50 -o p -- python3 $ROOT/tools/band_bench.py --workload 50m --ranks 8 --steps 10 > $ROOT/gpurun_out/r03q/trace_band50.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03q/trace_band10 -o p -- python3 $ROOT/tools/band_bench.py --workload 10m --ranks 8 --steps 10 > $ROOT/gpurun_out/r03q/trace_band10.log 2>&1
find $ROOT/gpurun_out/r03q -name "*kernel_trace.csv" -delete
echo ok
