#!/bin/bash
# Build libgs3d_hip.so from another git revision into build/ab/libgs3d_hip_<rev>.so (travels to the GPU
# box; load it with GS3D_LIB=build/ab/libgs3d_hip_<rev>.so for same-box A/B runs of two kernel versions).
#   tools/build_rev.sh <rev>
set -e
rev=${1:?usage: tools/build_rev.sh <rev>}
root=$(cd "$(dirname "$0")/.." && pwd)
short=$(git -C "$root" rev-parse --short "$rev")
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
git -C "$root" archive "$rev" wgpu-3dgs-core_amd/csrc include | tar -x -C "$tmp"
python3 - "$tmp" <<'PY'
import sys, os
csrc = os.path.join(sys.argv[1], "wgpu-3dgs-core_amd", "csrc")
text = open(os.path.join(csrc, "gs_kernel_lib.h")).read()
open(os.path.join(csrc, "_gen_kernel_lib_src.h"), "w").write(
    '// generated\nstatic const char k_kernel_lib_src[] = R"GSLIB(' + text + ')GSLIB";\n')
PY
mkdir -p "$root/build/ab"
out="$root/build/ab/libgs3d_hip_$short.so"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function \
    -I"$tmp/include" -o "$out" "$tmp/wgpu-3dgs-core_amd/csrc/gs3d.hip" "$tmp/wgpu-3dgs-core_amd/csrc/gs_ply.cpp" \
    "$tmp/wgpu-3dgs-core_amd/csrc/gs_spz.cpp" -lhiprtc -lz
echo "$out"
