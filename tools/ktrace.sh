#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace of one workload, single stream, and the per-kernel table
# (calls, average us) — the quick look between two edits; tools/profile.sh adds the PMC passes.
# usage: tools/ktrace.sh <tag> <workload> [steps] [extra bench args...]
set -o pipefail
TAG=${1:-t}
WL=${2:-1m}
STEPS=${3:-30}
shift 3 2>/dev/null
OUT=$PWD/gpurun_out/kt_${TAG}_${WL}
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 $PWD/bench.py --workload $WL --no-roofline --no-cpu-baseline --extra-workloads= --steps $STEPS --warmup 10 --timing-steps 0 --frames-in-flight 1 --no-steady --frame-samples 0 $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.log" 2>&1 || { tail -20 "$OUT/trace.log"; exit 1; }
cd - > /dev/null
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
frames = max(int(r["Calls"]) for r in rows if "k_preprocess" in r["Name"])      # once per frame whatever the rounds
tot = 0.0
print("frames traced: %d" % frames)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    calls = int(r["Calls"])
    if calls < frames:
        continue
    per = calls / frames
    avg = float(r["AverageNs"]) / 1e3
    tot += avg * per
    print("%-92s x%-4.2g %8.2f us" % (r["Name"].replace("void gs::", "")[:92], per, avg))
print("sum of kernel time per frame: %.1f us" % tot)
PY
find "$OUT" -name "*.csv" -size +2M -delete
tail -1 "$OUT/trace.log" | cut -c1-600
