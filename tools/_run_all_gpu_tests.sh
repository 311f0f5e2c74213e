set -e
mkdir -p gpurun_out/r03q
SECONDS=0
timeout -k 10 1100 python -m pytest tests -m gpu -x -v 2>&1 | grep --line-buffered -E "PASSED|FAILED|ERROR|passed|failed|Error|assert" | cut -c1-200 | tee gpurun_out/r03q/tests_all.log | grep --line-buffered -v PASSED || true
tail -2 gpurun_out/r03q/tests_all.log
echo "wall $SECONDS s"
