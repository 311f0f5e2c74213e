#!/bin/bash
# Runs on the GPU box: what one count of SQ_ACTIVE_INST_VALU is worth.
# tools/mb/mb_valu --pmc launches two kernels whose VALU-busy fraction is known by construction (a
# saturating stream of independent v_fma_f32, resp. v_pk_fma_f32, at 8 waves per SIMD); their
# counters give (SIMD-cycles of the launch) / (count) = the cycles one count stands for, which
# bench.py uses instead of a literal.  Output: gpurun_out/profiles_out/<tag>_valu_calibration.{txt,json}
set -o pipefail
TAG=${1:-r03}
REPO=$PWD
OUT=$REPO/gpurun_out/profiles_out
mkdir -p "$OUT" "$REPO/gpurun_out/valu_cal"
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o tools/mb/mb_valu tools/mb/mb_valu.hip || exit 1
tools/mb/mb_valu --pmc > "$OUT/${TAG}_valu_untraced.txt" || exit 1
cd /tmp
for SET in "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU"; do
  N=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --output-format csv -d "$REPO/gpurun_out/valu_cal/$N" -o pmc -- $REPO/tools/mb/mb_valu --pmc > "$REPO/gpurun_out/valu_cal/$N.log" 2>&1 || { tail -20 "$REPO/gpurun_out/valu_cal/$N.log"; }
done
cd - > /dev/null
python3 - "$REPO" "$TAG" <<'PY'
import csv, glob, json, os, re, sys
from collections import defaultdict
repo, tag = sys.argv[1], sys.argv[2]
out = os.path.join(repo, "gpurun_out", "profiles_out")
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(repo, "gpurun_out", "valu_cal", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"k<(\d+)>", row["Kernel_Name"])
        if m:
            acc[int(m.group(1))][row["Counter_Name"]].append(float(row["Counter_Value"]))
times = defaultdict(list)
for log in glob.glob(os.path.join(repo, "gpurun_out", "valu_cal", "*.log")) + [os.path.join(out, "%s_valu_untraced.txt" % tag)]:
    under = not log.endswith("untraced.txt")
    for line in open(log):
        m = re.match(r"CAL (\S+) kind=(\d+) launch=(\d+) ms=([0-9.]+) body_insts_per_simd=(\d+)", line)
        if m and int(m.group(3)) > 0:
            times[(int(m.group(2)), under)].append(float(m.group(4)))
SIMDS = 1024
res = {}
lines = ["VALU counter calibration (tools/calibrate_valu.sh): kernels with a known VALU-busy fraction of 1.0",
         "(independent instructions, 16 chains per thread, 8 waves per SIMD, 2048 workgroups of 256 threads)", ""]
for kind, name in ((0, "v_fma_f32"), (1, "v_pk_fma_f32")):
    c = {k: sum(v) / len(v) for k, v in acc.get(kind, {}).items()}
    if not c:
        continue
    ms_u = sum(times[(kind, False)]) / max(len(times[(kind, False)]), 1)
    ms_p = sum(times[(kind, True)]) / max(len(times[(kind, True)]), 1)
    body = 8 * 16384 * 16
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0                   # cycles of the launch (counter = sum over 8 XCDs)
    simd_cycles = gui * SIMDS
    r = dict(kernel="k<%d> (%s)" % (kind, name), counters=c, ms_untraced=ms_u, ms_under_pmc=ms_p,
             body_valu_insts_per_simd=body, ns_per_inst_untraced=ms_u * 1e6 / body,
             clock_ghz_from_gui_active=gui / (ms_p * 1e6) if ms_p else None,
             insts_counted_per_simd=c.get("SQ_INSTS_VALU", 0) / SIMDS,
             active_counts_per_inst=c.get("SQ_ACTIVE_INST_VALU", 0) / max(c.get("SQ_INSTS_VALU", 1), 1),
             simd_cycles_per_active_count=simd_cycles / max(c.get("SQ_ACTIVE_INST_VALU", 1), 1),
             simd_cycles_per_inst=simd_cycles / max(c.get("SQ_INSTS_VALU", 1), 1))
    res[name] = r
    lines.append("%s: %s" % (name, json.dumps(r)))
json.dump(res, open(os.path.join(out, "%s_valu_calibration.json" % tag), "w"), indent=1)
open(os.path.join(out, "%s_valu_calibration.txt" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
