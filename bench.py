#!/usr/bin/env python3
"""bench.py — headline benchmark of the render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one frame (repack-if-dirty -> preprocess -> scan/compact -> depth sort -> pair
expansion -> tile sort -> ranges -> blend [-> RCCL all-gather of the RGBA rows when N > 1]) of a
synthetic random-Gaussian
scene at 1920x1080 with the scene already resident in HBM.  The headline `value` is BASELINE.json's
metric, Msplats/s = Gaussians / frame time, on configs[1] (1 M Gaussians, SH degree 0); the
`roofline` object is measured on configs[2] (10 M Gaussians, SH degree 3, 224-byte records), the
configuration BASELINE.json quotes the HBM-read roofline on.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (N, sh config, cov config, sh_deg, payload bytes per Gaussian, W, H)
    "1m": dict(n=1_000_000, sh=3, cov=0, sh_deg=0, payload=44, width=1920, height=1080,
               label="1M synthetic Gaussians, SH degree 0 (ShNone/RotScale 48 B), 1920x1080"),
    "10m": dict(n=10_000_000, sh=0, cov=0, sh_deg=3, payload=224, width=1920, height=1080,
                label="10M synthetic Gaussians, SH degree 3 (ShSingle/RotScale 224 B), 1920x1080"),
    "10m-4k": dict(n=10_000_000, sh=0, cov=0, sh_deg=3, payload=224, width=3840, height=2160,
                   label="10M synthetic Gaussians, SH degree 3, 3840x2160"),
    "50m": dict(n=50_000_000, sh=1, cov=0, sh_deg=3, payload=134, width=1920, height=1080,
                label="50M synthetic Gaussians, fp16 SH (ShHalf/RotScale 144 B), 1920x1080"),
    "10m-deg0": dict(n=10_000_000, sh=0, cov=0, sh_deg=0, payload=224, width=1920, height=1080,
                     label="10M synthetic Gaussians, 224-B records but SH degree 0 (diagnostic)"),
    "100k": dict(n=100_000, sh=3, cov=0, sh_deg=0, payload=44, width=1920, height=1080,
                 label="100k synthetic Gaussians, SH degree 0 (debug size)"),
}


def upload_scene(gs, synth, dev, stream, wl, chunk=1_000_000):
    pod = gs.GaussianPod(wl["sh"], wl["cov"])
    buf = gs.GaussiansBuffer.new_empty(dev, pod, wl["n"])
    for first in range(0, wl["n"], chunk):
        cnt = min(chunk, wl["n"] - first)
        g = synth.scene(cnt, first=first)
        buf.update_range_with_pod(stream, first, pod.from_gaussian(g))
    return pod, buf


def run_workload(gs, synth, torch, dist, dev, stream, rank, world, wl, steps, warmup, timing_steps):
    """Returns dict with ms_per_frame (max over ranks), stats, per-stage ms (rank 0)."""
    from importlib import import_module
    par = import_module("wgpu_3dgs_core_amd.parallel")
    W, H = wl["width"], wl["height"]
    pod, buf = upload_scene(gs, synth, dev, stream, wl)
    cam = gs.camera_look_at((0, 0, 0), (0, 0, -1), (0, 1, 0), float(np.deg2rad(60.0)), W, H, 0.1, 100.0)
    gt = gs.gaussian_transform_pod(sh_deg=wl["sh_deg"])
    mt = gs.model_transform_pod()
    frame = par.allocate_frame(torch, H, W, world, "cuda")
    _, bands, _ = par.band_plan(H, world)
    band = bands[rank]
    r = gs.Renderer(dev)

    def step():
        r.render(stream, buf, gt, mt, cam, frame.data_ptr(), band=band, check=False)
        par.gather_frame(dist, frame, rank, world, H)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # separate short run with HIP-event stage timing on the launch stream
    r.set_timing(True)
    r.reset_stats()
    for _ in range(timing_steps):
        step()
    sync_all()
    st = r.stats()
    stages = {name: st.stage_ms[i] / max(st.timed_frames, 1) for i, name in enumerate(gs.STAGE_NAMES)}
    checksum = float(frame[:H].double().sum().item())
    out = dict(ms_per_frame=dt * 1e3 / steps, visible=int(st.visible), pairs=int(st.pairs),
               sort_passes=int(st.sort_passes), stages_ms=stages, checksum=checksum,
               timed_frames=int(st.timed_frames))
    r.destroy()
    buf.destroy()
    del frame
    return out


def stage_rooflines(wl, res):
    """Algorithmic bytes per launch over the event-timed stage duration.  The
    key/sort figure uses the survey's formulation (64-bit keys, 6 passes) as the common yardstick:
    `keys_sort` = scan + depth sort + expansion + tile sort of this implementation."""
    n, d, v = wl["n"], res["pairs"], res["visible"]
    px = wl["width"] * wl["height"]
    st = res["stages_ms"]
    tiles = ((wl["width"] + 15) // 16) * ((wl["height"] + 15) // 16)
    nominal_passes = -(-(32 + max(tiles - 1, 1).bit_length()) // 8)
    alg = {
        "preprocess": (n * wl["payload"], st["preprocess"]),                   # B_pre_read
        "keys_sort": (d * 12 + nominal_passes * d * 12 * 2,                    # B_key + B_sort
                      st["scan"] + st["depth_sort"] + st["expand"] + st["tile_sort"]),
        "blend": (d * (4 + 48) + px * 16, st["blend"]),                        # B_blend_read + B_out
    }
    out = {}
    for k, (b, ms) in alg.items():
        gbs = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out[k] = dict(bytes=b, ms=ms, achieved_gbs=gbs, frac=gbs / HBM_PEAK_GBS)
    total = sum(b for b, _ in alg.values()) + v * 48 + n * 4
    ms = st["frame"]
    gbs = total / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    out["frame"] = dict(bytes=total, ms=ms, achieved_gbs=gbs, frac=gbs / HBM_PEAK_GBS)
    return out


def cpu_baseline(wl, frames):
    """The CPU oracle (a restatement of the reference's conventions, NOT the reference binary —
    the Rust/WGSL reference cannot run here) timed on this host's cores on the same workload."""
    import synth
    from oracle import binding as ob
    ob.build()
    threads = ob.lib().gso_get_max_threads()
    g = synth.scene(wl["n"])
    pods = ob.pack(wl["sh"], wl["cov"], g)
    cam = ob.camera_look_at((0, 0, 0), (0, 0, -1), (0, 1, 0), float(np.deg2rad(60.0)), wl["width"],
                            wl["height"], 0.1, 100.0)
    gt, mt = ob.gaussian_transform(sh_deg=wl["sh_deg"]), ob.model_transform()
    times = []
    for _ in range(frames):
        t0 = time.perf_counter()
        ob.render(wl["sh"], wl["cov"], pods, gt, mt, cam, want_image=True)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return dict(value=wl["n"] / med / 1e6, unit="Msplats/s", cores=int(threads), kind="port",
                ms_per_frame=med * 1e3,
                sample="%d whole frames of workload '%s' (oracle/gs_oracle.c, OpenMP, median)" % (
                    frames, wl["label"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="1m", choices=sorted(WORKLOADS))
    ap.add_argument("--roofline-workload", default="10m", choices=sorted(WORKLOADS))
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=3)
    ap.add_argument("--timing-steps", type=int, default=20)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--force-device", type=int, default=-1, help="rehearsal: put every rank on this GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run\n"
                             % (args.gpus, world))
        sys.exit(2)
    if not torch.cuda.is_available():
        sys.stderr.write("bench.py: no GPU visible; the HIP path has no CPU fallback\n")
        sys.exit(3)
    if args.force_device >= 0:
        local = args.force_device
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    import synth
    import wgpu_3dgs_core_amd as gs
    dev = gs.Device(local)
    # launch on torch's current stream so the RCCL all-gather is ordered behind the blend kernel
    stream = dev.wrap_stream(torch.cuda.current_stream().cuda_stream)

    wl = WORKLOADS[args.workload]
    res = run_workload(gs, synth, torch, dist, dev, stream, rank, world, wl, args.steps, args.warmup,
                       args.timing_steps)
    roof_wl, roof = None, None
    if not args.no_roofline:
        roof_wl = WORKLOADS[args.roofline_workload]
        rsteps = max(5, min(args.steps, 20))
        roof = run_workload(gs, synth, torch, dist, dev, stream, rank, world, roof_wl, rsteps,
                            max(2, min(args.warmup, 5)), args.timing_steps)

    if rank == 0:
        value = wl["n"] / (res["ms_per_frame"] * 1e-3) / 1e6
        line = {
            "metric": "Msplats/s @1080p (Gaussians per second through proj+sort+blend)",
            "value": value,
            "unit": "Msplats/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": res["ms_per_frame"],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl["label"], "gaussians": wl["n"], "visible": res["visible"],
                       "pairs": res["pairs"], "sort_passes": res["sort_passes"],
                       "parallelism": "tile-row bands x%d + RCCL all-gather" % world if world > 1
                       else "single GPU", "image_checksum": res["checksum"]},
            "stages_ms": res["stages_ms"],
            "stage_rooflines": stage_rooflines(wl, res),
        }
        if roof is not None:
            sr = stage_rooflines(roof_wl, roof)
            pre = sr["preprocess"]
            traffic, fetched = None, None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    j = json.load(open(pmc))
                    traffic = j.get("preprocess_%s_bytes_per_launch" % args.roofline_workload)
                    fetched = j.get("preprocess_%s_fetch_bytes" % args.roofline_workload)
                except Exception:
                    traffic, fetched = None, None
            line["roofline"] = {
                "bound": "hbm", "kernel": "k_preprocess_banded<ShSingle,RotScale>",
                "workload": roof_wl["label"],
                "achieved": pre["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": pre["frac"], "traffic": traffic,
                "algorithmic_bytes_per_launch": pre["bytes"], "avg_launch_ms": pre["ms"],
                # SH bytes of Gaussians culled before SH evaluation may be skipped; the
                # fraction is computed on the bytes REQUIRED, the bytes FETCHED (rocprofv3 PMC) beside it
                "fetched_bytes_per_launch": fetched,
                "fetched_over_required": (fetched / pre["bytes"]) if fetched else None,
            }
            line["roofline_workload"] = {
                "value": roof_wl["n"] / (roof["ms_per_frame"] * 1e-3) / 1e6, "unit": "Msplats/s",
                "ms_per_step": roof["ms_per_frame"], "visible": roof["visible"], "pairs": roof["pairs"],
                "stages_ms": roof["stages_ms"], "stage_rooflines": sr,
            }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(wl, args.cpu_frames)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
