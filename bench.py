#!/usr/bin/env python3
"""bench.py — headline benchmark of the render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one frame (repack-if-dirty -> preprocess -> depth sort (its first pass compacts the
visible Gaussians) -> pair expansion -> tile sort -> ranges -> blend [-> RCCL all-gather of the RGBA
bands when N > 1]) of a synthetic random-Gaussian scene with the scene already resident in HBM; the
frames are enqueued back to back (gs_render_frame does not block in steady state).  The headline
`value` is BASELINE.json's metric, Msplats/s = Gaussians / frame time, on configs[1] (1 M Gaussians,
SH degree 0, 1080p).  The `roofline` object is measured on configs[2] (10 M Gaussians, SH degree 3,
224-byte records, 1080p), the configuration BASELINE.json quotes the HBM-read roofline on;
`roofline_nocull` is the same scene seen from a camera that culls nothing.  `workloads` carries
configs[3] (10 M at 3840x2160) and configs[4] (50 M, fp16 SH).  Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
SIMDS, CLOCK_GHZ = 1024, 2.4   # 256 CUs x 4 SIMDs, max clock (MI355X_MICROARCH.md chip table)
BLEND_LOOP_PK_SHARE = 0.50     # v_pk_* share of the VALU instructions of k_blend_grouped<0,4>'s two loop bodies (ISA: 38 of 76, rare finished-pixel blocks left out)

DEFAULT_EYE = (0.0, 0.0, 0.0)
WORKLOADS = {
    # payload = algorithmic bytes per Gaussian the preprocess stage must read
    "1m": dict(n=1_000_000, sh=3, cov=0, sh_deg=0, payload=44, width=1920, height=1080,
               label="1M synthetic Gaussians, SH degree 0 (ShNone/RotScale 48 B), 1920x1080"),
    "10m": dict(n=10_000_000, sh=0, cov=0, sh_deg=3, payload=224, width=1920, height=1080,
                label="10M synthetic Gaussians, SH degree 3 (ShSingle/RotScale 224 B), 1920x1080"),
    # the same scene from 12.5 units further back: every Gaussian is inside the frustum (V = N), so
    # nothing can be skipped by culling: bytes fetched = bytes required
    "10m-nocull": dict(n=10_000_000, sh=0, cov=0, sh_deg=3, payload=224, width=1920, height=1080,
                       eye=(0.0, 0.0, 12.5),
                       label="10M synthetic Gaussians, SH degree 3, 1920x1080, camera pulled back to z=+12.5 (V = N)"),
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


def par_nccl_options(dist):
    """ProcessGroupNCCL.Options with the collective's stream on the high-priority queues (None if this torch has none)"""
    try:
        opts = dist.ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
        return opts
    except Exception:
        return None


def _camera(gs, wl):
    eye = wl.get("eye", DEFAULT_EYE)
    return gs.camera_look_at(eye, (eye[0], eye[1], eye[2] - 1.0), (0, 1, 0), float(np.deg2rad(60.0)),
                             wl["width"], wl["height"], 0.1, 100.0)


def run_workload(gs, synth, torch, dist, dev, stream, rank, world, wl, steps, warmup, timing_steps,
                 frame_samples, rebalance=True, frames_in_flight=0, steady_frames=0):
    """Times `steps` pipelined frames between barriers (max over ranks), then `frame_samples`
    individually event-timed frames (median / min / p95), then a short run with HIP-event stage
    timing.  Returns a dict (rank-0 view; per-rank numbers where world > 1)."""
    from importlib import import_module
    par = import_module("wgpu_3dgs_core_amd.parallel")
    W, H = wl["width"], wl["height"]
    pod, buf = upload_scene(gs, synth, dev, stream, wl)
    cam = _camera(gs, wl)
    gt = gs.gaussian_transform_pod(sh_deg=wl["sh_deg"])
    mt = gs.model_transform_pod()
    plan = par.BandPlan(H, world)
    gbuf = par.allocate_gather(torch, plan, W, "cuda")
    r = gs.Renderer(dev)
    tiles_x, tiles_y = (W + 15) // 16, (H + 15) // 16

    def render(check=False):
        return r.render(stream, buf, gt, mt, cam, par.band_target_ptr(gbuf, plan, rank, W), band=plan.bands[rank],
                        check=check)

    # N > 1: frames go through a two-buffer pipeline — the all-gather of frame i (on RCCL's stream)
    # overlaps the rendering of frame i + 1; every frame is exchanged and assembled inside the timed region
    pipe = [None]
    count = [0]
    band_flags = [None]          # [G] int32 on the device: OR of every finished frame's per-band flags

    def finish(i):
        # the flags word of every band travelled with the band: OR them up (one tiny kernel, no host
        # round trip); a non-zero word = some rank skipped a frame, which voids the measurement
        img = pipe[0].finish(i)
        torch.bitwise_or(band_flags[0], pipe[0].flags(i), out=band_flags[0])
        return img

    # N > 1, frames in flight inside the rank: a band is a chain of small dependent kernels, so the rank's frames are
    # taken in turn by renderers on streams of different priority (parallel.lanes; 1 M / 8 ranks on one GPU: 0.162 ->
    # 0.076 ms per band with three, `tools/band_bench.py --frames-in-flight`); begin / render / submit run on the
    # lane's stream, the exchange is ordered behind it, finish on the main stream
    rank_lanes = [None]

    def step():
        if world == 1:
            render()
            return gbuf
        i = count[0]
        count[0] += 1
        if rank_lanes[0]:
            rr, gs_s, ts = rank_lanes[0][i % len(rank_lanes[0])]
            with torch.cuda.stream(ts):
                rr.render(gs_s, buf, gt, mt, cam, pipe[0].begin(i, rr), band=plan.bands[rank], check=False)
                pipe[0].submit(i)
        else:
            r.render(stream, buf, gt, mt, cam, pipe[0].begin(i, r), band=plan.bands[rank], check=False)
            pipe[0].submit(i)
        return finish(i - 1) if i else None

    def sync_all():
        if world > 1:
            if pipe[0] is not None and count[0]:
                finish(count[0] - 1)               # the last frame's exchange belongs to the region that ends here
            dist.barrier()
        torch.cuda.synchronize()

    plan_kind = "floor(g*R/G) tile rows"
    if world > 1 and rebalance:
        # one calibration frame: every rank measures the pairs per tile row of its own band, the
        # counts are summed over the ranks and the bands are re-cut to equal cost
        render(check=True)
        rows = par.row_costs_from_ranges(r.download_ranges(tiles_x * tiles_y), tiles_x, tiles_y, fixed=0.0)
        t = torch.from_numpy(rows).cuda()
        dist.all_reduce(t)
        plan = plan.rebalanced(t.cpu().numpy() + 64.0 * tiles_x)
        gbuf = par.allocate_gather(torch, plan, W, "cuda")
        plan_kind = "tile rows re-cut to equal pairs (one calibration frame)"
    fr = render(check=True)          # sizes the pair buffers for this band (blocking once)
    # ... and a few frames the validated way: the renderer picks its sorts, its rounds and their bounds per frame from the
    # reports of finished frames, which a viewer's loop has after a handful of frames; the W warm-up frames
    # that follow are enqueued without waiting and would otherwise start the timed region in the middle of that
    SETTLE = 6
    for _ in range(SETTLE):
        render(check=True)
    if world > 1:
        nl = max(1, min(int(frames_in_flight or 1), 2))
        if nl > 1:
            rank_lanes[0] = par.lanes(torch, gs, dev, nl, first_renderer=r)
            for rr, gs_s, ts in rank_lanes[0][1:]:      # the sizing frame of every further renderer (blocking once each), and its settling
                for _ in range(1 + SETTLE):
                    rr.render(gs_s, buf, gt, mt, cam, par.band_target_ptr(gbuf, plan, rank, W), band=plan.bands[rank], check=True)
            torch.cuda.synchronize()
        pipe[0] = par.FramePipeline(torch, dist, plan, rank, W, "cuda", depth=nl + 1)
        band_flags[0] = torch.zeros(world, dtype=torch.int32, device="cuda")
    # Frames in flight (single GPU): F renderers (each its own scratch buffers) on F streams of F different
    # priorities take the frames in turn, so that the latency-bound sort chain of frame i + 1 runs under the
    # VALU-bound blend of frame i.  Different priorities because HIP gives each priority level its own hardware
    # queues while streams of one priority may share a queue — under torch they do, and then nothing overlaps
    # (kernel trace: one Queue_Id for both streams; 0.327 ms per frame with or without the second stream, 0.275
    # with two priorities).  The contract's W warm-up + K timed steps are run this way FIRST, straight after the
    # sizing frames like the single-stream region used to be (so that the two are not told apart by how long the GPU
    # has been busy); every frame is a whole frame, bit-identical to the single-stream one (checked at the end).
    ring, dt2 = None, None
    if world == 1 and frames_in_flight and frames_in_flight > 1:
        F = int(frames_in_flight)
        ring = gs.FrameRing(dev, F)
        targets = [par.band_target_ptr(gbuf, plan, rank, W)]
        extra = [gs.Buffer(dev, size=W * H * 16) for _ in range(F - 1)]
        targets += [b.device_ptr() for b in extra]
        for k in range(F):               # the sizing frame of every lane's renderer (blocking once each)
            ring.render(buf, gt, mt, cam, targets[k], check=True)
        frame_no = [0]

        def pstep():
            frame_no[0] += 1
            ring.render(buf, gt, mt, cam, targets[(frame_no[0] - 1) % F])

        for _ in range(warmup):
            pstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pstep()
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0

        def finish_in_flight():
            psteady = None
            if steady_frames:
                for _ in range(100):
                    pstep()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steady_frames):
                    pstep()
                torch.cuda.synchronize()
                dts = time.perf_counter() - t0
                psteady = dict(frames=steady_frames, ms_per_step=dts * 1e3 / steady_frames,
                               value=wl["n"] / (dts / steady_frames) / 1e6, unit="Msplats/s")
            for _ in range(F):               # every lane's last frame is a frame of this camera
                pstep()
            torch.cuda.synchronize()
            flags = [int(fr_.flags) for fr_ in ring.wait()]
            if any(flags):
                raise RuntimeError("bench.py: a pipelined frame was skipped or flagged (per-lane flags %s)" % flags)
            ref = gbuf[:H].cpu().numpy().view(np.uint32)
            equal = all(np.array_equal(b.download(ring.streams[k + 1], np.float32).reshape(H, W, 4).view(np.uint32), ref)
                        for k, b in enumerate(extra))
            equal = equal and float(gbuf[:H].double().sum().item()) == checksum
            res = dict(frames_in_flight=F, ms_per_step=dt2 * 1e3 / steps, value=wl["n"] / (dt2 / steps) / 1e6,
                       unit="Msplats/s", stream_priorities=ring.priorities, steady_state=psteady,
                       images_bit_identical=bool(equal), measured="first: W warm-up + K timed frames straight after the sizing frames",
                       note="%d renderers on %d streams of different priority (= different hardware queues) take the "
                            "frames in turn; W warm-up + K timed frames as for the single stream" % (F, F))
            ring.close()
            for b in extra:
                b.release()
            return res

    for _ in range(warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    rank_ms = dt * 1e3 / steps
    per_rank = [rank_ms]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [float(x.item()) * 1e3 / steps for x in allt]
        dt = max(float(x.item()) for x in allt)

    # Steady state (single GPU): the timed region above starts a few frames after an idle GPU and still sees the
    # clocks ramp (same-box fit of total time over K: ~14 us per frame more for the first ~50 frames, then 0.3075 ms
    # flat at 1 M); a viewer renders continuously.  Reported BESIDE `value`, never as `value`: 100 more untimed frames,
    # then 200 timed ones between two synchronisations.
    steady = None
    if world == 1 and steady_frames:
        for _ in range(100):
            step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steady_frames):
            step()
        sync_all()
        dts = time.perf_counter() - t0
        steady = dict(frames=steady_frames, untimed_frames_before=100 + steps + warmup, ms_per_step=dts * 1e3 / steady_frames,
                      value=wl["n"] / (dts / steady_frames) / 1e6, unit="Msplats/s",
                      note="the same frames after the clocks have settled; `value` is the contract's W warm-up + K timed steps")
    # per-frame distribution: event pairs on the launch stream around single frames
    samples = []
    if frame_samples:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(frame_samples)]
        for a, b in evs:
            a.record()
            step()
            b.record()
        sync_all()
        samples = sorted(a.elapsed_time(b) for a, b in evs)
    # render-only and gather-only times per rank (world > 1), pipelined
    render_ms = gather_ms = None
    if world > 1:
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            render()
        torch.cuda.synchronize()
        render_ms = (time.perf_counter() - t0) * 1e3 / steps
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            par.gather_bands(dist, gbuf, plan, rank)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t0) * 1e3 / steps
        t = torch.tensor([render_ms, gather_ms], dtype=torch.float64, device="cuda")
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        render_ms = [float(x[0].item()) for x in allt]
        gather_ms = [float(x[1].item()) for x in allt]
    # separate short run with HIP-event stage timing on the launch stream
    stages = {}
    st = r.stats()
    if timing_steps:
        r.set_timing(True)
        r.reset_stats()
        for _ in range(timing_steps):
            step()
        sync_all()
        st = r.stats()
        stages = {name: st.stage_ms[i] / max(st.timed_frames, 1) for i, name in enumerate(gs.STAGE_NAMES)}
    img = step()
    if world > 1:
        img = finish(count[0] - 1)               # the frame just submitted, exchanged and assembled
    sync_all()
    if world > 1:
        # the pipelined frame (lanes, three gather buffers, async exchange) against the same frame rendered and
        # exchanged synchronously on the main stream: bit for bit, on every rank
        img = img.clone()
        plain = par.render_sharded(dist, torch, gbuf, plan, rank, W,
                                   lambda band, ptr: r.render(stream, buf, gt, mt, cam, ptr, band=band, check=False),
                                   renderer=r, check=True)
        if plain is None or not torch.equal(plain[:H].view(torch.int32), img[:H].view(torch.int32)):
            raise RuntimeError("bench.py: the pipelined N-GPU frame differs from the synchronously exchanged one (rank %d)" % rank)
    # no frame of this run may have been skipped (pair capacity): on one GPU the renderer's last result says
    # so, on N the OR of the flags words that travelled with every band of every finished frame
    if world > 1:
        skipped_bands = [int(x) for x in band_flags[0].cpu().tolist()]
        for rr in [r] + [l[0] for l in (rank_lanes[0] or [])[1:]]:
            rr.set_frame_flags_target(None)
    else:
        skipped_bands = [int(r.wait_frame().flags)]
    if any(skipped_bands):
        raise RuntimeError("bench.py: a frame was skipped or flagged (per-band flags %s; bit 0/1 = pair capacity exceeded / "
                           "skipped, bit 2 = radix rank watchdog): the timed region did not render every frame" % skipped_bands)
    checksum = float(img[:H].double().sum().item())
    fr = r.wait_frame()          # the LAST frame's result: its launches are a steady-state frame's (two rounds take ~36, the sizing frame 20)
    visible, pairs = int(st.visible), int(st.pairs)
    if world > 1:   # totals over the bands
        t = torch.tensor([pairs], dtype=torch.float64, device="cuda")
        dist.all_reduce(t)
        pairs = int(t.item())
    in_flight = finish_in_flight() if ring is not None else None
    out = dict(ms_per_frame=dt * 1e3 / steps, visible=visible, pairs=pairs, sort_passes=int(st.sort_passes),
               in_flight_run=in_flight, steady_state=steady,
               stages_ms=stages, checksum=checksum, launches=int(fr.launches), pair_capacity=int(fr.pair_capacity),
               per_rank_ms=per_rank, render_ms_per_rank=render_ms, gather_ms_per_rank=gather_ms,
               bands=plan.bands, band_plan=plan_kind, skipped_band_flags=skipped_bands,
               rank_lanes=len(rank_lanes[0]) if rank_lanes[0] else 1)
    si = r.sort_info()      # what the renderer chose for this workload
    out["sort_info"] = dict(depth_msd=int(si.depth_msd), depth_bucket_max=int(si.depth_bucket_max),
                            bucket_capacity=int(si.bucket_capacity), tile_msd=int(si.tile_msd), tile_masks=int(si.tile_masks),
                            rounds=int(si.rounds), round1=int(si.round1), tiles_done=int(si.tiles_done),
                            partitioned=int(si.partitioned))
    if samples:
        out["frame_ms"] = dict(samples=len(samples), median=samples[len(samples) // 2], min=samples[0],
                               p95=samples[min(len(samples) - 1, int(0.95 * len(samples)))], max=samples[-1],
                               timer="hipEvent pairs around single frames on the launch stream")
    for rr, gs_s, _ in (rank_lanes[0] or [])[1:]:
        rr.destroy()
    r.destroy()
    for _, gs_s, _ in (rank_lanes[0] or []):
        gs_s.close()
    buf.destroy()
    del gbuf
    return out


def stage_models(wl, res):
    """Per stage: the bytes THIS implementation's kernels have to move, from the counts of the
    frame (N, V, D) — not a generic model — over the event-timed stage duration.  All are HBM
    read + write streams except the blend, which is VALU-bound and carries no HBM fraction."""
    n, d, v = wl["n"], res["pairs"], res["visible"]
    px = wl["width"] * wl["height"]
    st = res["stages_ms"]
    if not st:
        return {}
    tiles_x, tiles_y = (wl["width"] + 15) // 16, (wl["height"] + 15) // 16
    tiles = tiles_x * tiles_y
    tile_bits = max(tiles - 1, 1).bit_length()
    tpasses = -(-tile_bits // 8)
    tkey = 2 if tiles <= 65536 else 4
    dpasses = max(res["sort_passes"] - tpasses, 1)
    rb = 4 if tiles_x <= 256 and tiles_y <= 256 and tiles <= 32768 else 8     # packed tile rects
    rect_writers = v if wl["sh"] != 3 else n                    # the two-phase kernel masks the rects of culled lanes
    # depth sort: pass 0 reads the N dense keys twice and writes V x (4 + 4); pass p > 0 reads its keys for the
    # histogram, keys + slots for the scatter, writes keys + slots; the pass before the last writes 2-byte keys,
    # the last one reads them and writes the slots only
    chunk_hist = os.environ.get("GS3D_CHUNK_HIST", "1") != "0"
    # pass 0: the N dense keys once for the scatter; its histogram sums the chunks' 1 KB rows (1 B per slot) that the
    # preprocess kernel counted — or reads the N keys once more with GS3D_CHUNK_HIST=0
    depth = n * 4 + (n if chunk_hist else n * 4) + v * ((2 if dpasses == 2 else 4) + 4 if dpasses > 1 else 4)
    for p in range(1, dpasses):
        kin = 2 if p == dpasses - 1 else 4
        kout = 0 if p == dpasses - 1 else (2 if p == dpasses - 2 else 4)
        depth += v * kin + v * (kin + 4) + v * (kout + 4)
    depth_what = ("pass 0 reads the N dense keys once (scatter) + 1 B per slot of per-chunk histogram rows and writes V x (key + 4 B); each further "
                  "pass reads its keys (hist) and keys + slots (scatter) and writes them; 2-byte keys into the last pass")
    if (res.get("sort_info") or {}).get("depth_msd"):
        # MSD-first: the top-digit scatter (N dense keys + 2 B per slot of 1024-bin histogram rows in, V x 8 B out), then the
        # bucket kernel reads V x 8 B and writes the V slots; the low digits never leave the CU
        depth = n * 4 + n * 2 + v * 8 + v * 8 + v * 4
        depth_what = ("MSD-first: one compacting scatter on the top 10 bits (N dense keys + 2 B per slot of histogram rows in, V x 8 B out), "
                      "k_bucket_sort reads V x 8 B, sorts the low bits on the CU and writes the V slots; latency-bound at this size (4 launches)")
    models = {
        "preprocess": (n * wl["payload"] + v * 36 + n * 4 + rect_writers * rb + (n if chunk_hist else 0),
                       "read N x payload; write 36-B records of the V visible, 4-B keys of all N (+ 1 B per slot of digit histogram), %d-B rects of the %s"
                       % (rb, "V visible" if wl["sh"] != 3 else "N"),),
        "depth_sort": (depth, depth_what),
        "expand": (v * (4 + rb) + v * rb,
                   "k_expand_count: V x (4-B slot + %d-B rect gather) -> V x %d B (the pairs themselves are produced "
                   "by the first kernel of the tile sort); sector waste of the gather not modelled" % (rb, rb)),
        "tile_sort": (v * (4 + rb) + d * (tkey + 4) + d * (tkey + 4) * 2 + (tpasses - 1) * d * (tkey + (tkey + 4) * 2),
                      "k_pairs_emit: V x (4 + %d) B in, D x (key + 4 B) out, first histogram fused; first scatter: D x (key + "
                      "4 B) read and written; each further pass: D keys (hist) + D x (key + 4 B) read and written" % rb),
    }
    # tile ranges: the kernel that ran decides the model (gs3d.hip: search from a pair capacity of 8 M,
    # GS3D_RANGES_SEARCH=0/1 forces)
    env = os.environ.get("GS3D_RANGES_SEARCH")
    searched = (env != "0") if env in ("0", "1") else res.get("pair_capacity", 0) >= (8 << 20)
    if searched:
        models["ranges"] = (ranges_search_bytes(d, tiles, tkey),
                            "k_tile_ranges_search: per round the DISTINCT 64-B sectors the 2 x tiles half-waves probe "
                            "(32 probes each; early rounds share their sectors between tiles), summed over the "
                            "ceil(log32 D) dependent rounds, + tile ranges written; a latency-bound kernel")
    else:
        models["ranges"] = (d * tkey + tiles * 8, "k_tile_ranges: D keys read, tile ranges written")
    out = {}
    if (res.get("sort_info") or {}).get("rounds") == 2:
        # A two-round frame (DESIGN.md §4.2 "rounds"): `pairs` counts what both rounds emitted; the stage events bracket
        # round 1's expansion, tile sort and ranges, and "blend" holds round 1's blend + the whole of round 2.  The byte
        # models above describe one pass over V and D and do not apply; preprocess and depth sort are unchanged.
        si = res["sort_info"]
        if si.get("partitioned"):
            # each round's depth sort takes only its side of a depth threshold: the stage is the threshold + round 1's sort
            models.pop("depth_sort", None)
            out["depth_sort"] = dict(bound="two_rounds", ms=st["depth_sort"], note="threshold + the sort of round 1's %d Gaussians only "
                                     "(first pass streams the N dense keys); round 2's sort is part of `blend`" % si["round1"])
        for k in ("expand", "tile_sort", "ranges"):
            models.pop(k, None)
            out[k] = dict(bound="two_rounds", ms=st[k], note="round 1 only (the nearest %d visible Gaussians)" % si["round1"])
        for k, (b, what) in models.items():
            ms = st[k]
            gbs = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            out[k] = dict(bound="hbm", model_bytes=b, model=what, ms=ms, achieved_gbs=gbs, frac=gbs / HBM_PEAK_GBS)
        out["blend"] = dict(bound="two_rounds", ms=st["blend"], pairs=d, pixels=px,
                            note="round 1's blend (%d of %d tiles finished), the slot bits and the compaction of round 2, its "
                                 "expansion, tile sort and resumed blend" % (si["tiles_done"], tiles))
        return out
    in_blend = os.environ.get("GS3D_RANGES_IN_BLEND")
    if (in_blend == "1") if in_blend in ("0", "1") else tiles > 16384:     # gs3d.hip: the rule that picks the range path
        # the blend workgroups find their own range (blend_tile_range): there is no range kernel to price
        models.pop("ranges")
        out["ranges"] = dict(bound="fused", ms=st["ranges"], note="tile ranges are found by the blend workgroups themselves "
                             "(32-ary search per tile inside k_blend_grouped); no launch, no stage of its own")
    for k, (b, what) in models.items():
        ms = st[k]
        gbs = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out[k] = dict(bound="latency" if k == "ranges" and searched else "hbm", model_bytes=b, model=what, ms=ms,
                      achieved_gbs=gbs, frac=gbs / HBM_PEAK_GBS)
    out["blend"] = dict(bound="valu", ms=st["blend"], pairs=d, pixels=px,
                        note="VALU-issue bound (profiles/: SQ_ACTIVE_INST_VALU vs kernel time); its HBM traffic is "
                             "a few per cent of the frame's and is not priced against the HBM roofline")
    return out


def ranges_search_bytes(d, tiles, tkey):
    """Bytes k_tile_ranges_search has to fetch: each of the 2 x tiles half-waves narrows [0, D] by 32
    probes per round (step = len / 32 + 1, the next interval is step - 1 long).  Round r's probes of all
    tiles lie on a grid of at most 32^(r+1) points, so the distinct sectors of a round are bounded by the
    grid, by the probes themselves (2 x tiles x the sectors one half-wave's 32 probes span) and by the
    array; 64 bytes per sector."""
    total, length, r = 0, d, 0
    array_sectors = (d * tkey + 63) // 64
    while length > 0:
        step = length // 32 + 1
        per_half = min(32, (32 * step * tkey + 63) // 64 + 1)
        grid = 32 ** (r + 1) if step * tkey >= 64 else array_sectors
        total += min(2 * tiles * per_half, grid, array_sectors) * 64
        length = min(step - 1, length)
        r += 1
    return total + tiles * 8


def frame_bytes_object(wl_name, wl, res):
    """Whole-frame byte models over the frame time (SURVEY.md §8d asks for B_frame / t_frame beside
    Msplats/s).  Three figures: (1) the survey's textbook pipeline (64-bit keys, six 8-bit passes over
    12-byte pairs) as an EQUIVALENT rate — this implementation does not move those bytes; (2) the sum
    of the bytes this implementation's HBM-bound stages have to move (stage_models) plus the image —
    the blend's gathers are NOT in it: its tiles saturate early and the counters show it fetching tens
    of MB, not D x 40 B (round 2 counted those and overstated the fraction); (3) the HBM traffic the
    PMC counters measured for one whole frame (sum over every kernel of the frame, separate rocprofv3
    --pmc passes, profiles/pmc_traffic.json) — the physical figure."""
    n, d, v = wl["n"], res["pairs"], res["visible"]
    px = wl["width"] * wl["height"]
    tiles = ((wl["width"] + 15) // 16) * ((wl["height"] + 15) // 16)
    passes = -(-(32 + max(tiles - 1, 1).bit_length()) // 8)
    survey = n * wl["payload"] + (v * 48 + n * 4) + d * 12 + passes * d * 12 * 2 + d * (4 + 48) + px * 16
    t = res["ms_per_frame"] * 1e-3
    out = dict(survey_model_bytes=survey, survey_model="N x payload + V x 48 + N x 4 + D x 12 + %d passes x D x 24 + D x 52 + W x H x 16"
               % passes, survey_equivalent_gbs=survey / t / 1e9)
    sm = stage_models(wl, res)
    if sm:
        mine = sum(x["model_bytes"] for x in sm.values() if "model_bytes" in x) + px * 16
        out.update(implementation_model_bytes=mine, implementation_gbs=mine / t / 1e9,
                   implementation_frac=mine / t / 1e9 / HBM_PEAK_GBS,
                   implementation_model="sum of the HBM-bound stage_models + W x H x 16 (image); the blend's record "
                                        "gathers are excluded: measured at tens of MB per frame (PMC), not D x 40 B")
    pmc, why = pmc_record("frame_%s" % wl_name)
    if res.get("per_rank_ms") and len(res["per_rank_ms"]) > 1:
        pmc, why = None, "PMC traffic was collected for the single-GPU frame"
    if pmc:
        tot = pmc["fetch_bytes"] + pmc["write_bytes"]
        out.update(pmc_frame_traffic_bytes=tot, pmc_frame_fetch_bytes=pmc["fetch_bytes"], pmc_frame_write_bytes=pmc["write_bytes"],
                   pmc_frame_gbs=tot / t / 1e9, pmc_frame_frac=tot / t / 1e9 / HBM_PEAK_GBS,
                   pmc_note="sum of FETCH_SIZE (x2, gfx950) + WRITE_SIZE over every kernel of one frame / this run's frame time")
    else:
        out["pmc_note"] = why
    return out


def kernel_source_stamp():
    h = hashlib.sha256()
    for f in ("gs_render_kernels.h", "gs_kernel_lib.h"):
        h.update(open(os.path.join(ROOT, "wgpu-3dgs-core_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_record(key):
    """Counters collected by separate rocprofv3 --pmc passes (tools/profile.sh -> profiles/pmc_traffic.json).
    They are only quoted while the kernel sources are the ones that were profiled."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, "no profiles/pmc_traffic.json"
    try:
        j = json.load(open(path))
    except Exception as e:     # noqa: BLE001
        return None, "unreadable: %s" % e
    rec = j.get(key)
    if rec is None:
        return None, "no entry %r" % key
    if j.get("kernel_source_stamp") != kernel_source_stamp():
        return None, "stale: kernels changed since the PMC passes (stamp %s != %s)" % (
            j.get("kernel_source_stamp"), kernel_source_stamp())
    return rec, None


def roofline_object(wl_name, wl, res, world=1):
    """roofline of the preprocess kernel on `wl`: achieved = ALGORITHMIC bytes per launch (payload
    bytes x Gaussians per launch) / average launch duration (HIP events on the launch stream).
    world > 1 (rank 0's launch): a rank must read the 44 B of position, colour and covariance of EVERY
    Gaussian but the SH bytes only of the Gaussians its band shows ("pos + cov traffic stays
    N x 44 B - the non-scaling term"): algorithmic bytes = N x 44 + V_rank x (payload - 44); the PMC traffic
    figures (collected on one GPU, whole frame) do not apply and are left out."""
    ms = res["stages_ms"]["preprocess"]
    alg = wl["n"] * wl["payload"]
    if world > 1:
        alg = wl["n"] * min(44, wl["payload"]) + res["visible"] * max(wl["payload"] - 44, 0)
    gbs = alg / (ms * 1e-3) / 1e9
    pmc, why = pmc_record("preprocess_%s" % wl_name)
    if world > 1:
        pmc, why = None, "PMC traffic was collected for the single-GPU frame; this is rank 0's band of %d" % world
    obj = {
        "bound": "hbm",
        "kernel": "k_preprocess_banded<ShSingle,RotScale,pipelined,nt>" if wl["sh"] != 3 else "k_preprocess<ShNone,RotScale>",
        "workload": wl["label"],
        "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
        "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms,
        "visible": res["visible"], "gaussians": wl["n"],
        "traffic": None,
    }
    if pmc:
        fetched, written = pmc["fetch_bytes"], pmc["write_bytes"]
        obj["traffic"] = fetched + written
        obj["fetched_bytes_per_launch"] = fetched
        obj["written_bytes_per_launch"] = written
        obj["fetched_over_required"] = fetched / alg
        # the physical rates of the same launch: what HBM actually delivered
        obj["physical_read_gbs"] = fetched / (ms * 1e-3) / 1e9
        obj["physical_read_frac"] = obj["physical_read_gbs"] / HBM_PEAK_GBS
        obj["physical_traffic_gbs"] = (fetched + written) / (ms * 1e-3) / 1e9
        obj["physical_traffic_frac"] = obj["physical_traffic_gbs"] / HBM_PEAK_GBS
    else:
        obj["traffic_note"] = why
    return obj


def blend_valu_object(res):
    """VALU issue of the blend kernel, priced with CALIBRATED costs: round 2
    multiplied SQ_ACTIVE_INST_VALU by a literal 4 "quad-cycles"; the calibration run
    (tools/calibrate_valu.sh: saturating v_fma_f32 / v_pk_fma_f32 streams at 8 waves per SIMD under the
    same counters, profiles/*_valu_calibration.txt) shows the counter is ONE per VALU instruction
    (== SQ_INSTS_VALU) and that an instruction occupies the SIMD's issue for 2.5 (v_fma_f32) to 4.8
    (v_pk_fma_f32) cycles.  The busy fraction is therefore given as a bracket — every instruction priced
    as the cheapest / as the dearest of the two — and as an estimate for the kernel's static mix."""
    pmc, why = pmc_record("blend_1m")
    cal, why2 = pmc_record("valu_calibration")
    ms = res["stages_ms"].get("blend")
    obj = {"bound": "valu", "kernel": "k_blend_grouped<Splat,4>", "avg_launch_ms": ms, "pairs": res["pairs"]}
    if pmc and cal and ms:
        insts = pmc["SQ_INSTS_VALU"]
        # SIMD-cycles of the launch from the same PMC pass: GRBM_GUI_ACTIVE is the sum over the 8 XCDs
        simd_cycles = pmc["GRBM_GUI_ACTIVE"] / 8.0 * SIMDS if pmc.get("GRBM_GUI_ACTIVE") else ms * 1e-3 * CLOCK_GHZ * 1e9 * SIMDS
        c_fma = cal["v_fma_f32"]["simd_cycles_per_inst"]
        c_pk = cal["v_pk_fma_f32"]["simd_cycles_per_inst"]
        pk_share = BLEND_LOOP_PK_SHARE
        obj.update(valu_insts_per_launch=insts, salu_insts_per_launch=pmc.get("SQ_INSTS_SALU"),
                   lds_insts_per_launch=pmc.get("SQ_INSTS_LDS"),
                   valu_wave_insts_per_pair=insts / max(res["pairs"], 1),
                   simd_cycles_per_launch=simd_cycles,
                   calibrated_cycles_per_inst={"v_fma_f32": c_fma, "v_pk_fma_f32": c_pk},
                   valu_busy_frac_if_all_fma=insts * c_fma / simd_cycles,
                   valu_busy_frac_if_all_pk=insts * c_pk / simd_cycles,
                   valu_busy_frac_static_mix=insts * ((1 - pk_share) * c_fma + pk_share * c_pk) / simd_cycles,
                   static_pk_share_of_loop=pk_share,
                   valu_insts_per_simd_per_ns=insts / SIMDS / (ms * 1e6),
                   note="counter units calibrated against kernels of known VALU-busy fraction 1.0 (profiles/); "
                        "SQ_ACTIVE_INST_VALU counts instructions on gfx950, not quad-cycles")
        # the bracket closed (round 4): every class of the loop priced with ITS OWN stream (tools/blend_table.py)
        try:
            bi = json.load(open(os.path.join(ROOT, "profiles", "blend_issue.json")))
            obj.update(valu_busy_frac_priced_by_class=insts * bi["ns_per_valu_inst_loop_mix"] / (ms * 1e6 * SIMDS),
                       priced_ns_per_valu_inst=bi["ns_per_valu_inst_loop_mix"], priced_by=bi["source"])
        except (OSError, KeyError, ValueError):
            pass
    else:
        obj["note"] = why or why2
    return obj


def cpu_baseline(wl, frames):
    """The CPU oracle (a restatement of the reference's conventions, NOT the reference binary — the
    Rust/WGSL reference cannot run here) timed on this host on the same workload: OpenMP on the CPUs
    this process may really use (affinity mask cut by the cgroup quota — NOT the machine's hardware
    threads: round 3 ran 256 threads on a 16-CPU share and measured its own barrier spinning), median
    of `frames` frames, and one frame on a single thread; per-stage seconds of both and the per-stage
    best of the two."""
    import synth
    from oracle import binding as ob
    ob.build()
    L = ob.lib()
    hw = int(L.gso_get_max_threads())
    threads = max(1, min(int(L.gso_effective_threads()), hw))
    g = synth.scene(wl["n"])
    pods = ob.pack(wl["sh"], wl["cov"], g)
    eye = wl.get("eye", DEFAULT_EYE)
    cam = ob.camera_look_at(eye, (eye[0], eye[1], eye[2] - 1.0), (0, 1, 0), float(np.deg2rad(60.0)), wl["width"],
                            wl["height"], 0.1, 100.0)
    gt, mt = ob.gaussian_transform(sh_deg=wl["sh_deg"]), ob.model_transform()
    names = ["preprocess", "keys", "sort", "ranges", "blend"]

    def run(count):
        times, stages = [], None
        for _ in range(count):
            t0 = time.perf_counter()
            _, _, _, st = ob.render(wl["sh"], wl["cov"], pods, gt, mt, cam, want_image=True)
            times.append(time.perf_counter() - t0)
            stages = st
        return float(np.median(times)), dict(zip(names, [round(x, 4) for x in stages]))

    L.gso_set_threads(threads)
    med, st_omp = run(frames)
    L.gso_set_threads(1)
    one, st_one = run(1)
    L.gso_set_threads(threads)
    best = {k: min(st_omp[k], st_one[k]) for k in names}
    return dict(value=wl["n"] / med / 1e6, unit="Msplats/s", cores=threads, kind="port",
                ms_per_frame=med * 1e3, stage_seconds=st_omp, hardware_threads=hw,
                single_thread=dict(value=wl["n"] / one / 1e6, unit="Msplats/s", cores=1, ms_per_frame=one * 1e3,
                                   stage_seconds=st_one),
                per_stage_best=dict(stage_seconds=best, ms_per_frame=sum(best.values()) * 1e3,
                                    value=wl["n"] / max(sum(best.values()), 1e-9) / 1e6),
                sample="%d whole frames (median) on %d OpenMP threads (the CPUs of this process: affinity + cgroup quota; "
                       "the host has %d hardware threads) + 1 whole frame on 1 thread of workload '%s' "
                       "(oracle/gs_oracle.c: every stage OpenMP-parallel; the blend visits every (pixel, splat) "
                       "of a tile without culling)" % (frames, threads, hw, wl["label"]))


def _r(x, digits=5):
    """float -> `digits` significant digits (the compact line has a byte budget)"""
    if x is None or isinstance(x, (str, bool, int)):
        return x
    return float("%.*g" % (digits, x))


def compact_line(line, detail_path, limit=3900):
    """The record the driver keeps: the contract's keys + the few numbers a judge needs, <= 4 KB.
    Everything else goes to the detail file (`--detail-out`)."""
    c = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                              "scaling", "vs_baseline", "dtype", "data")}
    c["value"], c["ms_per_step"] = _r(c["value"], 6), _r(c["ms_per_step"], 6)
    cfg = line["config"]
    c["config"] = {k: cfg[k] for k in ("workload", "gaussians", "visible", "pairs", "launches_per_frame", "parallelism",
                                       "frames_in_flight", "renderer_scratch_copies") if k in cfg}
    if line.get("frame_ms"):
        c["frame_ms_median"] = _r(line["frame_ms"]["median"])
    c["stages_ms"] = {k: _r(v, 4) for k, v in (line.get("stages_ms") or {}).items() if v}
    ro = line.get("roofline")
    if ro:
        keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_nocull", "algorithmic_bytes_per_launch",
                "avg_launch_ms", "fetched_over_required", "physical_traffic_frac", "traffic_note", "note")
        c["roofline"] = {k: _r(ro[k]) for k in keep if k in ro}
        c["roofline"]["workload"] = "10m" if ro["gaussians"] == 10_000_000 else ro["workload"]
        rw = line.get("roofline_workload")
        if rw:
            c["roofline_workload"] = {"ms": _r(rw["ms_per_step"]), "Msplats/s": _r(rw["value"]),
                                      "launches": rw["launches_per_frame"],
                                      "stages_ms": {k: _r(v, 4) for k, v in rw["stages_ms"].items() if v},
                                      "stage_frac": {k: _r(v["frac"], 3) for k, v in rw["stage_models"].items() if "frac" in v},
                                      "pmc_frame_frac": _r(rw["frame_bytes"].get("pmc_frame_frac"), 3)}
    nc = line.get("roofline_nocull")
    if nc:
        c["roofline_nocull"] = {"frac": _r(nc["frac"]), "avg_launch_ms": _r(nc["avg_launch_ms"]), "ms": _r(nc["frame"]["ms_per_step"])}
    cb = line.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                             "ms_per_frame": _r(cb["ms_per_frame"]), "hardware_threads": cb["hardware_threads"],
                             "single_thread_value": _r(cb["single_thread"]["value"]),
                             "per_stage_best_value": _r(cb["per_stage_best"]["value"]),
                             "sample": "%s: median of whole frames, OpenMP on `cores` threads (CPU share of this process); "
                                       "oracle/gs_oracle.c, a port, not the reference binary" % line["config"]["workload"][:14]}
    wls = line.get("workloads")
    if wls:
        c["workloads"] = {k: {"ms": _r(v["ms_per_step"]), "Msplats/s": _r(v["value"]),
                              "preprocess_read_frac": _r(v.get("preprocess_read_frac"), 3),
                              "stages_ms": {a: _r(b, 4) for a, b in v["stages_ms"].items() if b}} for k, v in wls.items()}
    di = line.get("distributed")
    if di:
        c["distributed"] = {k: ([_r(x, 4) for x in v] if isinstance(v, list) and v and isinstance(v[0], float) else v)
                            for k, v in di.items()}
        if wls:
            for k, v in wls.items():
                c["workloads"][k]["per_rank_ms"] = [_r(x, 4) for x in v.get("per_rank_ms", [])]
    ss = line.get("single_stream")
    if ss and line.get("in_flight_run"):    # `value` ran with frames in flight: the one-stream figure of the same W + K
        c["single_stream"] = {"ms_per_step": _r(ss["ms_per_step"], 6), "Msplats/s": _r(ss["value"])}
        if ss.get("steady_state"):
            c["single_stream"]["steady_ms"] = _r(ss["steady_state"]["ms_per_step"])
    if line.get("steady_state"):     # beside `value`: the same frames once the clocks have settled (200 frames)
        c["steady_state"] = {"ms_per_step": _r(line["steady_state"]["ms_per_step"]), "Msplats/s": _r(line["steady_state"]["value"]),
                             "frames": line["steady_state"]["frames"]}
    if (line.get("blend") or {}).get("valu_busy_frac_priced_by_class") is not None:
        c["blend_valu_issue_frac"] = _r(line["blend"]["valu_busy_frac_priced_by_class"], 3)   # VALU issue, priced class by class
    hr = line.get("hip_runtime") or {}
    c["hip"] = {"runtime": hr.get("runtime_version"), "compiled": hr.get("compiled_version"), "source": hr.get("source")}
    c["detail"] = detail_path
    # the byte budget is a contract with the driver's tail buffer: drop the optional parts, largest first
    for victim in ("workloads.stages_ms", "roofline_workload.stages_ms", "stages_ms", "distributed", "workloads"):
        if len(json.dumps(c)) <= limit:
            break
        if "." in victim:
            a, b = victim.split(".")
            if a == "workloads":
                for v in c.get(a, {}).values():
                    v.pop(b, None)
            else:
                c.get(a, {}).pop(b, None)
        else:
            c.pop(victim, None)
    return c


def self_launch(args_list, n):
    """`python3 bench.py --gpus N` typed by hand (no torchrun around it): start the N ranks as a CHILD
    process tree — before this process has touched the GPU, and never by exec — relay rank 0's lines
    and exit with the child's code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + args_list
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["GS3D_BENCH_SELF_LAUNCHED"] = "1"
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    last = None
    for ln in proc.stdout:
        if ln.strip():
            last = ln.rstrip("\n")
    rc = proc.wait()
    if last is not None:
        print(last, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="1m", choices=sorted(WORKLOADS))
    ap.add_argument("--roofline-workload", default="10m", choices=sorted(WORKLOADS))
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extra-workloads", default="10m-4k,50m",
                    help="comma-separated workloads also run and reported under `workloads` ('' = none)")
    ap.add_argument("--cpu-frames", type=int, default=3)
    ap.add_argument("--timing-steps", type=int, default=20)
    ap.add_argument("--frame-samples", type=int, default=100)
    ap.add_argument("--no-in-flight", action="store_true", help="single stream only (same as --frames-in-flight 1)")
    ap.add_argument("--frames-in-flight", type=int, default=3,
                    help="single GPU: renderers / priority streams that take the headline's frames in turn (1 = one stream)")
    ap.add_argument("--no-steady", action="store_true", help="skip the steady-state measurement (200 frames after 100 more untimed ones)")
    ap.add_argument("--no-rebalance", action="store_true", help="N > 1: keep the floor(g*R/G) band plan")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--force-device", type=int, default=-1, help="rehearsal: put every rank on this GPU")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="file that receives the full record (every stage model, frame distribution, PMC figures); "
                         "stdout carries ONE compact JSON line (<= 4 KB) as its last line")
    ap.add_argument("--detail-stdout", action="store_true", help="also print the full record (before the compact line)")
    ap.add_argument("--launch-check", action="store_true",
                    help="only the launch plumbing (works without a GPU): every rank joins the process group, one "
                         "all_reduce, rank 0 prints a JSON line with the backend and the world size")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # typed by hand: become the launcher (nothing in this process has touched the GPU yet)
        sys.exit(self_launch(sys.argv[1:], args.gpus))

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
    if args.launch_check:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo" if args.backend != "nccl" or not torch.cuda.is_available() else "nccl")
            t = torch.ones(1)
            if dist.get_backend() == "nccl":
                torch.cuda.set_device(local if args.force_device < 0 else args.force_device)
                t = t.cuda()
            dist.all_reduce(t)
            ok, be, ws = int(t.item()) == world, dist.get_backend(), dist.get_world_size()
            dist.barrier()
            dist.destroy_process_group()
        else:
            ok, be, ws = True, None, 1
        if rank == 0:
            print(json.dumps({"launch_check": bool(ok), "n_gpus": args.gpus, "distributed": {"backend": be, "world_size": ws},
                              "self_launched": os.environ.get("GS3D_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
        sys.exit(0 if ok else 4)
    if not torch.cuda.is_available():
        sys.stderr.write("bench.py: no GPU visible; the HIP path has no CPU fallback\n")
        sys.exit(3)
    if args.force_device >= 0:
        local = args.force_device
    torch.cuda.set_device(local)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            # RCCL's stream on the HIGH-priority queues: the frames are rendered on a stream of the default priority,
            # and two streams of one priority may share a hardware queue — the all-gather of frame i
            # would then sit in front of the kernels of frame i + 1 instead of running beside them
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), pg_options=par_nccl_options(dist))
        else:
            dist.init_process_group(args.backend)
        backend = dist.get_backend()

    import synth
    import wgpu_3dgs_core_amd as gs
    from importlib import import_module
    hiprt = import_module("wgpu_3dgs_core_amd._hiprt")
    hiprt.check("torch")       # one HIP runtime for torch, RCCL and the product library
    # torchrun exports OMP_NUM_THREADS=1: the scene generator (OpenMP) would take minutes per rank for
    # the 50 M scene.  Give every rank its share of the host's cores instead.
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    synth.set_threads(max(1, cores // max(world, 1)))
    dev = gs.Device(local)
    # A dedicated NON-default stream, made torch's current stream: RCCL orders a collective behind the
    # current stream and work.wait() makes the current stream wait for it, so the all-gather of frame i
    # overlaps the rendering of frame i + 1 (the legacy null stream would serialise with RCCL's stream)
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    assert tstream.cuda_stream != 0
    stream = dev.wrap_stream(tstream.cuda_stream)

    def run(name, steps, warmup, samples, frames_in_flight=0, steady_frames=0):
        return run_workload(gs, synth, torch, dist, dev, stream, rank, world, WORKLOADS[name], steps, warmup,
                            args.timing_steps, samples, rebalance=not args.no_rebalance, frames_in_flight=frames_in_flight,
                            steady_frames=steady_frames)

    wl = WORKLOADS[args.workload]
    res = run(args.workload, args.steps, args.warmup, args.frame_samples,
              frames_in_flight=0 if args.no_in_flight else args.frames_in_flight,
              steady_frames=0 if args.no_steady else 200)
    rsteps, rwarm = max(5, min(args.steps, 20)), max(2, min(args.warmup, 5))
    roof = nocull = None
    if not args.no_roofline:
        roof = run(args.roofline_workload, rsteps, rwarm, args.frame_samples)
        if args.roofline_workload == "10m" and world == 1:
            nocull = run("10m-nocull", rsteps, rwarm, 0)
    extras = {}
    for name in [x for x in args.extra_workloads.split(",") if x]:
        if name in (args.workload, args.roofline_workload) or args.no_roofline:
            continue
        extras[name] = run(name, min(rsteps, 10), rwarm, 20)

    if rank == 0:
        def summary(w, rr, name=None):
            d = {"workload": w["label"], "value": w["n"] / (rr["ms_per_frame"] * 1e-3) / 1e6, "unit": "Msplats/s",
                 "ms_per_step": rr["ms_per_frame"], "frame_ms": rr.get("frame_ms"), "visible": rr["visible"],
                 "pairs": rr["pairs"], "launches_per_frame": rr["launches"], "sort_info": rr.get("sort_info"), "stages_ms": rr["stages_ms"],
                 "stage_models": stage_models(w, rr), "frame_bytes": frame_bytes_object(name, w, rr)}
            if rr["stages_ms"]:
                pre = rr["stages_ms"]["preprocess"]
                alg = w["n"] * w["payload"] if world == 1 else w["n"] * min(44, w["payload"]) + rr["visible"] * max(w["payload"] - 44, 0)
                d["preprocess_read_frac"] = alg / (pre * 1e-3) / 1e9 / HBM_PEAK_GBS    # world > 1: rank 0's band (roofline_object)
            if world > 1:
                d.update(per_rank_ms=rr["per_rank_ms"], per_rank_ms_min=min(rr["per_rank_ms"]),
                         per_rank_ms_max=max(rr["per_rank_ms"]), render_ms_per_rank=rr["render_ms_per_rank"],
                         gather_ms_per_rank=rr["gather_ms_per_rank"], bands=rr["bands"], band_plan=rr["band_plan"])
            return d

        # single GPU: the headline is measured with the frames in flight (run_workload), the single-stream
        # figure of the same W + K steps stays beside it
        fl = res.get("in_flight_run")
        if fl and not fl["images_bit_identical"]:
            raise RuntimeError("bench.py: the pipelined frames differ from the single-stream frame")
        head_ms = fl["ms_per_step"] if fl else res["ms_per_frame"]
        value = wl["n"] / (head_ms * 1e-3) / 1e6
        line = {
            # what `value` times is in the metric's own name: with F > 1 frames in flight it is the THROUGHPUT of
            # F renderers taking the frames in turn, not one frame at a time; `single_stream` beside it is the latter
            "metric": "Msplats/s @1080p (Gaussians per second through proj+sort+blend%s)" % (
                "; %d frames in flight on %d priority streams" % (fl["frames_in_flight"], fl["frames_in_flight"]) if fl and world == 1 else ""),
            "value": value,
            "unit": "Msplats/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head_ms,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": wl["label"], "gaussians": wl["n"], "visible": res["visible"],
                       "pairs": res["pairs"], "sort_passes": res["sort_passes"], "sort_info": res.get("sort_info"),
                       "launches_per_frame": res["launches"],
                       "parallelism": "tile-row bands x%d + one %s all-gather (%s), %d frames in flight per rank" % (
                           world, "RCCL" if backend == "nccl" else "%s (rehearsal, host-staged)" % backend, res["band_plan"],
                           res.get("rank_lanes", 1))
                       if world > 1 else ("single GPU, %d frames in flight on %d priority streams" % (
                           fl["frames_in_flight"], fl["frames_in_flight"]) if fl else "single GPU, one stream"),
                       "frames_in_flight": fl["frames_in_flight"] if fl else res.get("rank_lanes", 1),
                       # every lane is a renderer of its own: per-slot outputs, sort and pair buffers once per lane
                       "renderer_scratch_copies": fl["frames_in_flight"] if fl else res.get("rank_lanes", 1),
                       "image_checksum": res["checksum"]},
            "single_stream": {"ms_per_step": res["ms_per_frame"], "value": wl["n"] / (res["ms_per_frame"] * 1e-3) / 1e6,
                              "unit": "Msplats/s", "steady_state": res.get("steady_state"),
                              "note": "the same W + K frames on one stream, one frame at a time on the device"},
            "frame_ms": res.get("frame_ms"),
            "stages_ms": res["stages_ms"],
            "stage_models": stage_models(wl, res),
            "frame_bytes": frame_bytes_object(args.workload, wl, res),
            "in_flight_run": res.get("in_flight_run"),
            "steady_state": (fl or {}).get("steady_state") or res.get("steady_state"),
            "blend": blend_valu_object(res) if args.workload == "1m" and world == 1 else None,
            "hip_runtime": {"source": hiprt.info()["source"], "libamdhip64": hiprt.mapped()["libamdhip64"],
                            "compiled_version": gs.hip_versions()[0], "runtime_version": gs.hip_versions()[1],
                            "driver_version": gs.hip_versions()[2],
                            "launch_stream": "dedicated non-default stream (torch.cuda.Stream, current)"},
        }
        if world > 1:
            line["distributed"] = {"backend": backend, "world_size": dist.get_world_size(),
                                   "per_rank_ms": res["per_rank_ms"], "per_rank_ms_min": min(res["per_rank_ms"]),
                                   "per_rank_ms_max": max(res["per_rank_ms"]),
                                   "render_ms_per_rank": res["render_ms_per_rank"],
                                   "gather_ms_per_rank": res["gather_ms_per_rank"], "bands": res["bands"],
                                   "band_plan": res["band_plan"], "skipped_band_flags": res["skipped_band_flags"]}
        if roof is not None:
            roof_wl = WORKLOADS[args.roofline_workload]
            line["roofline"] = roofline_object(args.roofline_workload, roof_wl, roof, world)
            if world > 1:
                line["roofline"]["note"] = ("rank 0's band: algorithmic bytes = N x 44 B (geometry of every Gaussian) + "
                                            "V_rank x 180 B (SH of the band's visible Gaussians)")
            line["roofline_workload"] = summary(roof_wl, roof, args.roofline_workload)
            if nocull is not None:
                line["roofline_nocull"] = roofline_object("10m-nocull", WORKLOADS["10m-nocull"], nocull)
                line["roofline_nocull"]["frame"] = summary(WORKLOADS["10m-nocull"], nocull, "10m-nocull")
                # the fraction that does not lean on culling, next to the headline one
                line["roofline"]["frac_nocull"] = line["roofline_nocull"]["frac"]
                line["roofline"]["achieved_nocull"] = line["roofline_nocull"]["achieved"]
        if extras:
            line["workloads"] = {k: summary(WORKLOADS[k], v, k) for k, v in extras.items()}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(wl, args.cpu_frames)
        detail_path = None
        try:
            os.makedirs(os.path.dirname(os.path.abspath(args.detail_out)), exist_ok=True)
            with open(args.detail_out, "w") as fh:
                json.dump(line, fh, indent=1)
            detail_path = os.path.relpath(args.detail_out, ROOT)
        except OSError as e:
            sys.stderr.write("bench.py: could not write %s: %s\n" % (args.detail_out, e))
        if args.detail_stdout:
            print(json.dumps(line), flush=True)
        print(json.dumps(compact_line(line, detail_path)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
