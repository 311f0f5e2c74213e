#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes for one workload.
# usage: tools/profile.sh <tag> <workload> [steps]
set -o pipefail
TAG=${1:-r01}
WL=${2:-10m}
STEPS=${3:-5}
OUT=$PWD/gpurun_out/prof_${TAG}_${WL}
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 $PWD/bench.py --workload $WL --no-roofline --no-cpu-baseline --steps $STEPS --warmup 2 --timing-steps 0 --frames-in-flight 1"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.log" 2>&1 || { tail -20 "$OUT/trace.log"; exit 1; }
# The counter passes render the warm-up + timed frames only (no latency samples, no steady-state run): with those, a process
# queues ~400 frames back to back, and from ~16 000 queued dispatches on (409 two-round partitioned frames of 47 launches)
# rocprofv3's counter collection stopped making progress in one run out of two (round 5; the same command without --pmc, or
# with a wait per frame, never did).  A pass that still stalls is cut off and reported instead of silencing the whole round.
CMD_PMC="$CMD --frame-samples 0 --no-steady"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- $CMD_PMC > "$OUT/pmc_fetch.log" 2>&1 || { echo "FETCH_SIZE pass failed or stalled"; tail -20 "$OUT/pmc_fetch.log"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- $CMD_PMC > "$OUT/pmc_write.log" 2>&1 || { echo "WRITE_SIZE pass failed or stalled"; tail -20 "$OUT/pmc_write.log"; exit 1; }
cd - > /dev/null
python3 tools/profile_summary.py "$OUT" "$TAG" "$WL" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep the merged-back payload small: drop the raw per-dispatch CSVs once summarised
find "$OUT" -name "*.csv" -size +2M -delete
