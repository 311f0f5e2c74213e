#!/bin/bash
# Runs on the GPU box (via gpurun): everything profiles/ holds for one round.
#   1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command (the line the driver records)
#   2. per workload (1m, 10m, 10m-nocull): kernel trace + separate --pmc FETCH_SIZE / WRITE_SIZE passes
#   3. SQ counters of the blend kernel at 1 M (VALU / SALU / LDS instruction counts, busy cycles)
#   4. gpurun_out/profiles_out/pmc_traffic.json, stamped with the hash of the kernel sources
# usage: tools/profile_round.sh <tag>      (PMC passes never share a run with --kernel-trace)
set -o pipefail
TAG=${1:-r02}
REPO=$PWD
OUT=$REPO/gpurun_out/profiles_out
mkdir -p "$OUT"
export TMPDIR=/tmp

bash tools/calibrate_valu.sh "$TAG" > "$OUT/${TAG}_valu_calibration.log" 2>&1 || { tail -5 "$OUT/${TAG}_valu_calibration.log"; exit 1; }

for WL in 1m 10m 10m-nocull; do
  bash tools/profile.sh "$TAG" "$WL" 7 > "$OUT/${TAG}_${WL}_profile.log" 2>&1 || { tail -5 "$OUT/${TAG}_${WL}_profile.log"; exit 1; }
done

SQ1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
bash tools/pmc.sh "${TAG}_sq1" 1m "$SQ1" 5 > "$OUT/${TAG}_1m_sq1.txt" 2>&1 || { tail -5 "$OUT/${TAG}_1m_sq1.txt"; exit 1; }
bash tools/pmc.sh "${TAG}_sq2" 1m "$SQ2" 5 > "$OUT/${TAG}_1m_sq2.txt" 2>&1 || { tail -5 "$OUT/${TAG}_1m_sq2.txt"; exit 1; }

python3 - "$OUT" "$TAG" <<'PY'
import hashlib, json, os, re, sys
out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(out))
rec = {}
for wl in ("1m", "10m", "10m-nocull"):
    rows = json.load(open(os.path.join(out, "%s_%s_summary.json" % (tag, wl))))
    for r in rows:
        if r["kernel"].startswith("k_preprocess") and r["fetch_bytes_per_launch"] is not None:
            rec["preprocess_%s" % wl] = dict(kernel=r["kernel"], avg_us=r["avg_us"], fetch_bytes=r["fetch_bytes_per_launch"],
                                             write_bytes=r["write_bytes_per_launch"])
            break
# whole-frame HBM traffic: every kernel of one frame, fetch + write per launch x launches per frame
for wl in ("1m", "10m", "10m-nocull"):
    rows = json.load(open(os.path.join(out, "%s_%s_summary.json" % (tag, wl))))
    frames = max([r["calls"] for r in rows if r["kernel"].startswith("k_preprocess")] or [0])      # (once per frame whatever the rounds)
    if not frames:
        continue
    fetch = write = 0.0
    per_kernel = {}
    for r in rows:
        if not r["kernel"].startswith("k_") or r["fetch_bytes_per_launch"] is None or r["calls"] < frames:
            continue          # upload-time kernels (repack, morton, ...) run once, not per frame
        per = r["calls"] / frames
        fetch += r["fetch_bytes_per_launch"] * per
        write += r["write_bytes_per_launch"] * per
        per_kernel[r["kernel"][:60]] = dict(launches_per_frame=per, avg_us=r["avg_us"], fetch_bytes=r["fetch_bytes_per_launch"],
                                            write_bytes=r["write_bytes_per_launch"])
    rec["frame_%s" % wl] = dict(fetch_bytes=fetch, write_bytes=write, frames=frames, kernels=per_kernel)
cal_path = os.path.join(out, "%s_valu_calibration.json" % tag)
if os.path.exists(cal_path):
    rec["valu_calibration"] = json.load(open(cal_path))
blend = {}
for f in ("%s_1m_sq1.txt" % tag, "%s_1m_sq2.txt" % tag):
    for line in open(os.path.join(out, f)):
        if line.startswith("k_blend"):
            for m in re.finditer(r"(\w+)=([0-9.e+-]+)", line):
                blend[m.group(1)] = float(m.group(2))
            blend["kernel"] = line.split()[0]
rec["blend_1m"] = blend
h = hashlib.sha256()
for f in ("gs_render_kernels.h", "gs_kernel_lib.h"):
    h.update(open(os.path.join(root, "wgpu-3dgs-core_amd", "csrc", f), "rb").read())
rec["kernel_source_stamp"] = h.hexdigest()[:16]
rec["source"] = ("rocprofv3 separate --pmc passes of `bench.py --workload <wl> --no-roofline` (tools/profile_round.sh %s): "
                 "FETCH_SIZE x2 (gfx950 counts a wide coalesced read at half its bytes) and KiB -> bytes; "
                 "WRITE_SIZE KiB -> bytes; SQ_* averaged per launch" % tag)
json.dump(rec, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
PY

# the default command, traced (the JSON line goes to bench.json; PMC-derived fields come from the file above)
cp "$OUT/pmc_traffic.json" "$REPO/profiles/pmc_traffic.json"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_default_trace" -o trace -- python3 $REPO/bench.py > "$OUT/${TAG}_bench_default_line.json" 2> "$OUT/${TAG}_bench_default.err" || { tail -20 "$OUT/${TAG}_bench_default.err"; exit 1; }
cd - > /dev/null
cp "$(find "$OUT/${TAG}_default_trace" -name "*kernel_stats.csv" | head -1)" "$OUT/${TAG}_bench_default_kernel_stats.csv"
cp "$(find "$OUT/${TAG}_default_trace" -name "*kernel_trace.csv" | head -1)" "$OUT/${TAG}_default_kernel_trace.csv"
python3 - "$OUT" "$TAG" <<'PY'
import csv, json, sys
out, tag = sys.argv[1], sys.argv[2]
b = json.loads([l for l in open("%s/%s_bench_default_line.json" % (out, tag)) if l.startswith("{")][-1])
pmc = json.load(open("%s/pmc_traffic.json" % out))
rows = list(csv.DictReader(open("%s/%s_bench_default_kernel_stats.csv" % (out, tag))))
# The SAME launches, seen by both timers: the traced default command runs the workloads 1m, 10m,
# 10m-nocull, 10m-4k, 50m in this order, each starting with one upload (k_repack_planar*); the
# k_preprocess_banded<0, 0> launches between two uploads belong to one workload.
trace = sorted(csv.DictReader(open("%s/%s_default_kernel_trace.csv" % (out, tag))), key=lambda r: int(r["Start_Timestamp"]))
groups, cur = [], None
for r in trace:
    n = r["Kernel_Name"]
    if "k_repack_planar" in n:
        cur = []
        groups.append(cur)
    elif cur is not None and "k_preprocess_banded<0, 0" in n:
        cur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
groups = [g for g in groups if g]
with open("%s/%s_bench_default_agreement.txt" % (out, tag), "w") as f:
    f.write("roofline kernel k_preprocess_banded<ShSingle, RotScale, pipelined, nt>, average launch duration of the SAME launches of the default command:\n")
    f.write("bench.py (HIP events on the launch stream, its timing run) vs rocprofv3 --kernel-trace (all launches of the workload)\n")
    for (key, wl), g in zip((("roofline", "10m"), ("roofline_nocull", "10m-nocull")), groups):
        if key in b:
            f.write("  %-11s bench %.4f ms   rocprofv3 %.4f ms over %d launches (min %.4f, max %.4f)\n" % (
                wl, b[key]["avg_launch_ms"], sum(g) / len(g), len(g), min(g), max(g)))
    f.write("other processes on the same box (--kernel-trace --stats of `bench.py --workload <wl> --no-roofline --frames-in-flight 1`, %s_<wl>_summary.md;\n" % tag)
    f.write("a fresh process places its buffers anew, which moves this HBM-bound kernel by a few per cent):\n")
    for key, wl in (("roofline", "10m"), ("roofline_nocull", "10m-nocull")):
        if ("preprocess_%s" % wl) in pmc:
            f.write("  %-11s rocprofv3 %.4f ms   (%s)\n" % (wl, pmc["preprocess_%s" % wl]["avg_us"] / 1e3, pmc["preprocess_%s" % wl]["kernel"]))
    f.write("kernel stats of the traced default command itself (one kernel name serves 10m, 10m-nocull and 10m-4k launches):\n")
    for r in rows:
        if "k_preprocess" in r["Name"]:
            f.write("  rocprofv3 %s: calls %s avg %.4f ms\n" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e6))
    f.write(json.dumps({k: b[k] for k in ("metric", "value", "unit", "ms_per_step", "frame_ms", "roofline", "roofline_nocull", "blend") if k in b}) + "\n")
print(open("%s/%s_bench_default_agreement.txt" % (out, tag)).read())
PY
find "$OUT" -name "*.csv" -size +2M -delete
rm -rf "$OUT/${TAG}_default_trace" "$OUT/${TAG}_default_kernel_trace.csv"
