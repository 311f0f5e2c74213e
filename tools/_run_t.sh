set -e
mkdir -p gpurun_out/r03q
timeout -k 10 1000 python -m pytest tests/test_gpu_render.py tests/test_gpu_switches.py tests/test_gpu_fullsize.py -m gpu -x -v 2>&1 | grep --line-buffered -E "PASSED|FAILED|ERROR|passed|failed|Error|assert" | cut -c1-200 | tee gpurun_out/r03q/tests_bl2.log | grep --line-buffered -v PASSED || true
tail -3 gpurun_out/r03q/tests_bl2.log
