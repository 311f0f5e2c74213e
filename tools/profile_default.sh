#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace of the DEFAULT bench command, i.e. the
# exact command whose JSON line the driver records.  usage: tools/profile_default.sh <tag>
set -o pipefail
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_${TAG}_default
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 $REPO/bench.py > "$OUT/bench.json" 2> "$OUT/trace.log" || { tail -20 "$OUT/trace.log"; exit 1; }
cd - > /dev/null
STATS=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
cp "$STATS" "$OUT/kernel_stats.csv"
python3 - "$OUT" <<'PY'
import csv, json, sys
out = sys.argv[1]
line = [l for l in open(out + "/bench.json") if l.startswith("{")][-1]
b = json.loads(line)
rows = list(csv.DictReader(open(out + "/kernel_stats.csv")))
pre = [r for r in rows if "k_preprocess" in r["Name"] and "<0, 0>" in r["Name"]]
print("bench roofline.avg_launch_ms = %.4f ms" % b["roofline"]["avg_launch_ms"])
for r in pre:
    print("rocprofv3 %s: calls %s avg %.4f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6))
print(json.dumps({k: b[k] for k in ("metric", "value", "unit", "ms_per_step", "roofline")}))
PY
find "$OUT/trace" -name "*.csv" -size +2M -delete
