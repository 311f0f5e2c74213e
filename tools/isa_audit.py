#!/usr/bin/env python3
"""Per kernel of libgs3d_hip.so: the ORDER in which hipcc emitted global loads (L), stores (S), global
atomics (A) and s_waitcnt vmcnt(n) (wn), barriers (|), plus VGPRs and LDS — no GPU needed.

A run of `L w0 L w0 L w0` where the source says "issue all loads, then use them" means the loads run
as DEPENDENT round trips (a run-time cache-policy flag had put a branch and an
s_waitcnt vmcnt(0) behind every chunk load of the preprocess kernel; the histogram kernels waited
for each 16-byte load before its four LDS atomics).  `L8 w7 w6 … w0` is what a batch looks like.

    python tools/isa_audit.py [substring of the demangled kernel name ...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "wgpu-3dgs-core_amd", "csrc", "gs3d.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only"]


def main():
    want = sys.argv[1:]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "gs3d.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, SRC], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    heads = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_ZN2gs\d+k_", l)]
    names = subprocess.run(["c++filt"], input="\n".join(h for _, h in heads), capture_output=True, text=True).stdout.split("\n")
    for (start, _), nm in zip(heads, names):
        nm = re.sub(r"\(.*", "", nm).replace("void gs::", "").replace("gs::", "")
        if want and not any(w in nm for w in want):
            continue
        end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
        seq = []
        for l in lines[start:end]:
            t = l.strip()
            if t.startswith("global_load"):
                seq.append("L")
            elif t.startswith("global_store"):
                seq.append("S")
            elif t.startswith("global_atomic"):
                seq.append("A")
            elif t.startswith("s_barrier"):
                seq.append("|")
            elif t.startswith("s_waitcnt") and "vmcnt" in t:
                seq.append("w" + re.search(r"vmcnt\((\d+)\)", t).group(1))
        comp, prev, cnt = [], None, 0
        for x in seq + [None]:
            if x == prev:
                cnt += 1
                continue
            if prev is not None:
                comp.append(prev + (str(cnt) if cnt > 1 and prev in "LSA" else ""))
            prev, cnt = x, 1
        meta = " ".join(m.strip("; ") for m in lines[end:end + 90] if re.search(r"NumVgprs:|LDSByteSize:|Occupancy:", m))
        print("%-64s %s\n%66s%s" % (nm[:64], " ".join(comp), "", meta))


if __name__ == "__main__":
    main()
