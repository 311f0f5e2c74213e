#!/bin/bash
# usage: tools/pmc_cmd.sh <tag> "<COUNTER1 COUNTER2 ...>" <python script + args ...>   (runs on the GPU box)
# Per-kernel averages of the counters over every launch of the given command (as tools/pmc.sh, for any tool).
set -o pipefail
TAG=$1; CTRS=$2; shift 2
OUT=$PWD/gpurun_out/pmc_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc" -o pmc -- python3 "$ROOT/$1" "${@:2}" > "$OUT/pmc.log" 2>&1 || { tail -20 "$OUT/pmc.log"; exit 1; }
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
        print(k[:64].ljust(64), "  ".join("%s=%.4g n=%d" % (c, sum(v) / len(v), len(v)) for c, v in sorted(d.items())))
PY
find "$OUT" -name "*.csv" -size +1M -delete
