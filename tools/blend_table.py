#!/usr/bin/env python3
"""Prices the inner loop of k_blend_grouped<Splat, 4> instruction class by instruction class
("close the blend's bracket or find its gap").

  1. hipcc -S of the product source; the kernel's inner loop (two blend steps per trip) is cut out of
     the ISA and its instructions are counted by class.  The blocks that only run when a pixel
     reaches T < 1e-4 in this very step (rare) are left out.
  2. every class is priced with ITS OWN calibration stream (tools/mb/mb_valu.hip: independent
     instructions of that class at 7 waves per SIMD — the blend's occupancy — in ns per wave-instruction
     per SIMD; `--valu-log` = the tool's output on the GPU box).
  3. the kernel's PMC counters (profiles/pmc_traffic.json `blend_1m`: SQ_INSTS_VALU / _SALU / _LDS,
     GRBM_GUI_ACTIVE) give the SIMD-cycles of the launch and the number of wave-steps.
Output: a markdown table (stdout) — per class: count per step, cost, issue cycles per step; the priced
sum per step against the measured SIMD-cycles per step.

    python tools/blend_table.py --valu-log gpurun_out/.../mb_valu.txt [--pmc profiles/pmc_traffic.json]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter, OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "wgpu-3dgs-core_amd", "csrc", "gs3d.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only",
         "-I" + os.path.join(ROOT, "include")]

# ISA mnemonic -> (class, calibration stream of tools/mb/mb_valu.hip)
CLASSES = OrderedDict([
    (r"v_pk_fma_f32", ("v_pk_fma_f32", "v_pk_fma_f32")),
    (r"v_pk_mul_f32", ("v_pk_mul_f32", "v_pk_mul_f32")),
    (r"v_pk_add_f32", ("v_pk_add_f32", "v_pk_add_f32")),
    (r"v_fma_f32|v_mul_f32|v_add_f32", ("v_mul/fma_f32", "v_mul_f32")),
    (r"v_sub_f32", ("v_sub_f32", "v_sub_f32")),
    (r"v_min_f32|v_max_f32", ("v_min_f32", "v_min_f32")),
    (r"v_cmp_\w+_e64", ("v_cmp -> sgpr pair", "v_cmp_ge_f32 -> sgpr pair")),
    (r"v_cmp_\w+", ("v_cmp -> vcc", "v_cmp_ge_f32 -> vcc")),
    (r"v_cndmask_b32", ("v_cndmask_b32", "v_cndmask_b32 (vcc)")),
    (r"v_lshl_add_u32|v_lshrrev_b32|v_lshlrev_b32|v_add_u32|v_add3_u32", ("v int (lshl_add ...)", "v_lshl_add_u32")),
    (r"v_and_b32|v_or_b32", ("v_and_b32", "v_and_b32")),
    (r"v_mov_b64|v_mov_b32", ("v_mov", "v_mov_b64")),
])


def kernel_isa():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "gs3d.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, SRC], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN2gs15k_blend_groupedILi0ELi4E"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    return lines[start:end]


def inner_loop(body):
    """basic blocks of the LAST depth-2 loop of the kernel (the blend loop; the first one fills the lists)"""
    heads = [i for i, l in enumerate(body) if "Inner Loop Header: Depth=2" in l]
    head = heads[-1]
    label = body[head - 1].split(":")[0].strip() if body[head - 1].strip().startswith(".LBB") else None
    if label is None:       # the comment sits on the label's own line in some versions
        label = body[head].split(":")[0].strip()
    # the loop = every line from the lowest label that jumps back to `label` ... simpler: all blocks marked
    # "in Loop: Header=<label> Depth=2" plus the header block itself
    tag = "Header=%s Depth=2" % label.lstrip(".L")
    blocks, cur, name, inside = [], [], None, False
    for i, l in enumerate(body):
        t = l.strip()
        is_label = t.startswith(".LBB") or t.startswith("; %bb.")
        if is_label:
            if cur and inside:
                blocks.append((name, cur))
            cur, name = [], t.split(":")[0]
            inside = (tag in t) or (i == head - 1) or (i == head) or (i + 1 < len(body) and i + 1 == head)
            continue
        if inside and t and not t.startswith(";"):
            cur.append(t)
    if cur and inside:
        blocks.append((name, cur))
    return label, blocks


def classify(mn):
    base = re.sub(r"_(e32|e64)$", "", mn)
    key = base + ("_e64" if mn.endswith("_e64") and base.startswith("v_cmp") else "")
    for pat, (cls, cal) in CLASSES.items():
        if re.fullmatch(pat, key):
            return cls, cal
    if mn.startswith("ds_"):
        return "LDS (ds_read)", None
    if mn == "s_nop":
        return "s_nop", None
    if mn.startswith("s_waitcnt"):
        return "s_waitcnt", None
    if mn.startswith("s_"):
        return "SALU", None
    if mn.startswith("v_"):
        return "v other: " + mn, "v_mul_f32"
    return "other: " + mn, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--valu-log", required=True, help="output of tools/mb/mb_valu on the GPU box")
    ap.add_argument("--pmc", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    ap.add_argument("--waves", default="7w")
    ap.add_argument("--json-out", default=os.path.join(ROOT, "profiles", "blend_issue.json"),
                    help="machine-readable summary for bench.py's `blend` object")
    args = ap.parse_args()

    cost = {}
    for line in open(args.valu_log):
        m = re.match(r"(.+?)\s{2,}1w:", line)
        if not m:
            continue
        name = m.group(1).strip()
        per = dict((k, float(v)) for k, v in re.findall(r"(\dw):\s+([0-9.]+) ns/inst", line))
        cost[name] = per
    body = kernel_isa()
    label, blocks = inner_loop(body)
    rare = [n for n, b in blocks if any(x.startswith("s_bcnt1") for x in b)]
    counts, cal_of = Counter(), {}
    for n, b in blocks:
        if n in rare:
            continue
        for ins in b:
            mn = ins.split()[0]
            cls, cal = classify(mn)
            counts[cls] += 1
            cal_of[cls] = cal
    steps_per_trip = 2.0
    pmc = json.load(open(args.pmc)).get("blend_1m", {}) if os.path.exists(args.pmc) else {}
    clock_ghz = None
    print("# k_blend_grouped<Splat, 4>: issue cost of the blend loop, class by class\n")
    print("Inner loop `%s` of the ISA hipcc emits for gfx950 (two blend steps per trip; the blocks that only run when a "
          "pixel reaches T < 1e-4 in that step are left out: %s).  Cost = ns per wave-instruction per SIMD of an "
          "independent stream of THAT class at %s per SIMD (tools/mb/mb_valu.hip, `%s`).\n" % (
              label, ", ".join(rare) or "none", args.waves, os.path.relpath(args.valu_log, ROOT)))
    print("| class | per trip | per step | ns / inst (%s) | ns per step |" % args.waves)
    print("|---|---|---|---|---|")
    total_ns, valu_per_step, missing = 0.0, 0.0, []
    for cls, c in sorted(counts.items(), key=lambda kv: -kv[1]):
        cal = cal_of.get(cls)
        ns = cost.get(cal, {}).get(args.waves) if cal else None
        if cls == "v_cndmask_b32":
            # the stream of v_cndmask alone reads a VCC nobody wrote and stalls on it (9.7 ns at every occupancy): the
            # select is priced from the compare + select pair instead: 2 x (ns per instruction of the pair) - v_cmp
            pair = cost.get("v_cmp_ge_f32 + v_cndmask (2)", {}).get(args.waves)
            cmp_ = cost.get("v_cmp_ge_f32 -> vcc", {}).get(args.waves)
            if pair is not None and cmp_ is not None:
                ns = max(2.0 * pair - cmp_, 0.0)
        per_step = c / steps_per_trip
        if cal:
            valu_per_step += per_step
        if cal and ns is None:
            missing.append(cal)
        sub = per_step * ns if ns is not None else None
        if sub is not None:
            total_ns += sub
        print("| %s | %d | %.1f | %s | %s |" % (cls, c, per_step, "%.2f" % ns if ns is not None else "—",
                                                 "%.1f" % sub if sub is not None else "—"))
    print("\nVALU instructions per step: %.1f; **priced VALU issue per step: %.1f ns** (per SIMD, i.e. per wave-step).\n" % (
        valu_per_step, total_ns))
    if missing:
        print("(no calibration stream found for: %s)\n" % sorted(set(missing)))
    if args.json_out:
        json.dump(dict(kernel="k_blend_grouped<Splat,4>", valu_insts_per_step=valu_per_step, priced_ns_per_step=total_ns,
                       ns_per_valu_inst_loop_mix=total_ns / valu_per_step, waves_per_simd=args.waves,
                       classes={k: v / steps_per_trip for k, v in counts.items()},
                       source="tools/blend_table.py: loop ISA counted by class, each class priced with its own stream of "
                              "tools/mb/mb_valu.hip (%s)" % os.path.relpath(args.valu_log, ROOT)),
                  open(args.json_out, "w"), indent=1)
    if pmc.get("SQ_INSTS_VALU") and pmc.get("GRBM_GUI_ACTIVE"):
        insts = pmc["SQ_INSTS_VALU"]
        gui = pmc["GRBM_GUI_ACTIVE"] / 8.0            # cycles of the launch (the counter sums the 8 XCDs)
        lds = pmc.get("SQ_INSTS_LDS", 0.0)
        ms = None
        for k, v in json.load(open(args.pmc)).get("frame_1m", {}).get("kernels", {}).items():
            if k.startswith("k_blend_grouped"):
                ms = v["avg_us"] * 1e-3
        wave_steps = insts / valu_per_step            # upper bound: staging instructions are charged as steps
        print("Measured (profiles/pmc_traffic.json `blend_1m`, one launch at 1 M): SQ_INSTS_VALU %.4g, SQ_INSTS_SALU %.4g, "
              "SQ_INSTS_LDS %.4g, GRBM_GUI_ACTIVE / 8 = %.4g cycles%s." % (
                  insts, pmc.get("SQ_INSTS_SALU", 0.0), lds, gui, " over %.4f ms" % ms if ms else ""))
        if ms:
            clock_ghz = gui / (ms * 1e6)
            simd_ns = ms * 1e6 * 1024.0          # SIMD-nanoseconds of the launch
            priced = insts * (total_ns / valu_per_step)
            print("\n* SIMD-time of the launch: %.4f ms x 1024 SIMDs = %.4g SIMD-ns (clock %.2f GHz from GRBM_GUI_ACTIVE)." % (
                ms, simd_ns, clock_ghz))
            print("* All %.4g VALU instructions priced at the loop's mix (%.2f ns each): %.4g SIMD-ns = **%.2f of the launch**." % (
                insts, total_ns / valu_per_step, priced, priced / simd_ns))
            print("* Wave-steps <= SQ_INSTS_VALU / %.1f = %.4g; measured SIMD-time per wave-step >= %.1f ns against %.1f ns of "
                  "priced VALU issue." % (valu_per_step, wave_steps, simd_ns / wave_steps, total_ns))
    return 0


if __name__ == "__main__":
    sys.exit(main())
