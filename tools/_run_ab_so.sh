set -e
mkdir -p gpurun_out/r03q
L=wgpu-3dgs-core_amd/lib/libgs3d_hip.so
for rep in 1 2; do
for v in base new; do
  cp tools/_ab/libgs3d_$v.so $L
  python tools/ab_bench.py --workloads 1m,10m,10m-4k --set $v: --steps 40 | tee -a gpurun_out/r03q/ab_blend.log
done
done
cp tools/_ab/libgs3d_new.so $L
python -m pytest tests/test_gpu_render.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r03q/tests_blend.log
