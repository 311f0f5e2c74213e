#!/bin/bash
# usage: tools/pmc.sh <tag> <workload> "<COUNTER1 COUNTER2 ...>" [steps]   (runs on the GPU box)
set -o pipefail
TAG=$1; WL=$2; CTRS=$3; STEPS=${4:-3}
OUT=$PWD/gpurun_out/pmc_${TAG}_${WL}
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 $PWD/bench.py --workload $WL --no-roofline --no-cpu-baseline --steps $STEPS --warmup 1 --timing-steps 0 --frames-in-flight 1"
cd /tmp
rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc" -o pmc -- $CMD > "$OUT/pmc.log" 2>&1 || { tail -20 "$OUT/pmc.log"; exit 1; }
cd - > /dev/null
python3 - "$OUT" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void gs::", "").replace("gs::", "").strip()
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(acc.items()):
    if k.startswith("k_"):
        print(k[:48].ljust(48), "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
find "$OUT" -name "*.csv" -size +1M -delete
