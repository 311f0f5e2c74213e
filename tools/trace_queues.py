#!/usr/bin/env python3
"""rocprofv3 kernel_trace.csv -> which hardware queue (Queue_Id) and stream each kernel family ran on.
    python3 tools/trace_queues.py <kernel_trace.csv>"""
import csv
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
fam = Counter()
for r in rows:
    n = r["Kernel_Name"]
    kind = "rccl" if "nccl" in n.lower() or "rccl" in n.lower() else "gs" if "gs::" in n else "other"
    fam[(kind, r["Queue_Id"], r.get("Stream_Id"))] += 1
for (kind, q, s), c in sorted(fam.items()):
    print("%-6s queue %-3s stream %-3s  %d kernels" % (kind, q, s, c))
