#!/usr/bin/env python3
"""A/B of environment switches on one box: runs bench.py once per (setting, workload), one process at
a time, and prints ms per frame and the stage table side by side.

    python tools/ab_bench.py --workloads 1m,10m,50m --set A: --set B:GS3D_FUSED_RECT_GATHER=0 [--repeat 2]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(env_kv, wl, steps):
    env = dict(os.environ)
    env.update(env_kv)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--no-roofline", "--no-cpu-baseline",
           "--no-in-flight", "--extra-workloads", "", "--steps", str(steps), "--warmup", "5", "--frame-samples", "50"]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if res.returncode != 0:
        print(res.stderr[-2000:])
        raise SystemExit(1)
    return json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="1m,10m")
    ap.add_argument("--set", action="append", default=[], help="NAME:VAR=VAL,VAR=VAL")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    sets = []
    for s in args.set or ["default:"]:
        name, _, kv = s.partition(":")
        sets.append((name, dict(x.split("=", 1) for x in kv.split(",") if x)))
    for wl in args.workloads.split(","):
        rows = []
        for rep in range(args.repeat):
            for name, kv in sets:
                j = run(kv, wl, args.steps)
                rows.append((name, j))
                st = j["stages_ms"]
                print("%-5s %-14s ms/step %.4f  median %.4f | %s" % (
                    wl, name, j["ms_per_step"], j.get("frame_ms_median") or (j.get("frame_ms") or {}).get("median", 0.0),
                    " ".join("%s %.4f" % (k[:6], v) for k, v in st.items() if k not in ("repack", "scan", "frame"))), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
