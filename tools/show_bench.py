#!/usr/bin/env python3
"""Pretty-print the JSON line of a bench.py log."""
import json
import sys
line = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]
d = json.loads(line)
print("value %.1f %s  ms/step %.4f  n_gpus %d" % (d["value"], d["unit"], d["ms_per_step"], d["n_gpus"]))
print("  stages", {k: round(v, 4) for k, v in d["stages_ms"].items()})
if "roofline_workload" in d:
    r = d["roofline_workload"]
    print("roofline workload: %.1f Msplats/s  ms/step %.4f" % (r["value"], r["ms_per_step"]))
    print("  stages", {k: round(v, 4) for k, v in r["stages_ms"].items()})
    print("  roofline", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["roofline"].items() if k in ("achieved", "frac", "avg_launch_ms", "traffic")})
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
