"""Child process of tests/test_gpu_rccl.py — NOT collected by pytest.

    python tests/rccl_child.py product-first|torch-first [frames]

Runs, in a FRESH process on one GPU, what `bench.py --gpus N` runs on N: the product library and
torch in the given import order (DESIGN.md §5 "One HIP runtime"), a real `nccl` (= RCCL) process
group of world size 1, a dedicated non-default torch stream wrapped as the launch stream, and
`parallel.FramePipeline` with the collective FORCED (in-place `all_gather_into_tensor`,
`async_op=True`, `work.wait()`), frames of different cameras in flight on the two gather buffers.
Every frame that comes out of the pipeline must equal, bit for bit, the same frame rendered plainly
into a product buffer on the product's own stream.  Prints one JSON line.
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402


def main():
    order = sys.argv[1]
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    if order == "product-first":
        import wgpu_3dgs_core_amd as gs
        dev = gs.Device(0)            # the product creates its HIP device before torch is even imported
        import torch
    elif order == "torch-first":
        import torch
        assert torch.cuda.is_available()
        torch.zeros(4, device="cuda:0").sum().item()      # torch's runtime is live before the product loads
        import wgpu_3dgs_core_amd as gs
        dev = gs.Device(0)
    else:
        raise SystemExit("order must be product-first or torch-first")
    import torch.distributed as dist
    from importlib import import_module
    par = import_module("wgpu_3dgs_core_amd.parallel")
    hiprt = import_module("wgpu_3dgs_core_amd._hiprt")
    import synth

    mapped = hiprt.check("torch (%s)" % order)          # exactly one libamdhip64 / libhsa-runtime64
    assert len(mapped["libamdhip64"]) == 1, mapped
    assert torch.cuda.is_available(), "torch sees no GPU next to the product library"
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    opts = dist.ProcessGroupNCCL.Options()       # as bench.py does: RCCL's stream on the high-priority queues, so that
    opts.is_high_priority_stream = True          # it cannot share a hardware queue with the render stream
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), pg_options=opts)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1

    W, H, N = 1280, 720, 200_000
    pod = gs.GaussianPodWithShSingleCov3dRotScaleConfigs
    own = dev.create_stream()
    buf = gs.GaussiansBuffer.new_with_pods(dev, pod, pod.from_gaussian(synth.scene(N)))
    gt, mt = gs.gaussian_transform_pod(sh_deg=3), gs.model_transform_pod()
    cams = [gs.camera_look_at((0.3 * i, 0.1 * i, 0.2 * i), (0.3 * i, 0.0, 0.2 * i - 1.0), (0, 1, 0),
                              float(np.deg2rad(60.0)), W, H, 0.1, 100.0) for i in range(frames)]
    # reference frames: plain renders on the product's own stream, no torch involved
    r = gs.Renderer(dev)
    img = gs.Buffer(dev, size=W * H * 16)
    want = []
    for cam in cams:
        r.render(own, buf, gt, mt, cam, img.device_ptr())
        want.append(hashlib.sha256(img.download(own, np.float32).tobytes()).hexdigest())
    assert len(set(want)) == frames, "the test frames must differ from each other"

    # the N > 1 path of bench.py on a dedicated (non-default) stream
    s = torch.cuda.Stream()
    assert s.cuda_stream != 0
    stream = dev.wrap_stream(s.cuda_stream)
    plan = par.BandPlan(H, 1)
    got, kinds = [], set()
    with torch.cuda.stream(s):
        pipe = par.FramePipeline(torch, dist, plan, 0, W, "cuda", force_collective=True)
        # poison the gather buffers: a frame the collective or the pipeline mixes up shows
        for b in pipe.bufs:
            b.fill_(float("nan"))
        for i, cam in enumerate(cams):
            r.render(stream, buf, gt, mt, cam, pipe.begin(i), band=plan.bands[0], check=False)
            pipe.submit(i)
            kinds.add(type(pipe.work[i % 2]).__name__)
            assert pipe.work[i % 2] is not None, "the forced collective returned no work handle"
            if i:
                got.append(pipe.finish(i - 1).clone())
        got.append(pipe.finish(frames - 1).clone())
        pipe.drain()
        # the synchronous form too (render_sharded: what a one-off frame uses)
        gbuf = par.allocate_gather(torch, plan, W, "cuda")
        gbuf.fill_(float("nan"))
        r.render(stream, buf, gt, mt, cams[0], par.band_target_ptr(gbuf, plan, 0, W), band=plan.bands[0], check=False)
        par.gather_bands(dist, gbuf, plan, 0, force=True)
        sync_img = par.assemble(torch, gbuf, plan).clone()
        # A band that overflows its pair capacity is skipped on the device (GS_ERR_PAIR_CAPACITY).  The
        # flags word travels inside the band's chunk through the same collective: finish(check=True)
        # must drop that frame (FramePipeline presented a torn image) and the next frame,
        # rendered with the grown buffers, must be exact.
        far = gs.camera_look_at((0.0, 0.0, 60.0), (0.0, 0.0, 0.0), (0, 1, 0), float(np.deg2rad(60.0)), W, H, 0.1, 100.0)
        gt_big = gs.gaussian_transform_pod(size=4.0, sh_deg=3)
        r2 = gs.Renderer(dev)
        pipe2 = par.FramePipeline(torch, dist, plan, 0, W, "cuda", force_collective=True)
        r2.render(stream, buf, gt, mt, far, pipe2.begin(0, r2), band=plan.bands[0], check=True)     # sizing frame: few pairs
        pipe2.submit(0)
        far_img = pipe2.finish(0, check=True)
        assert far_img is not None and int(pipe2.flags(0)[0].item()) == 0
        r2.render(stream, buf, gt_big, mt, cams[0], pipe2.begin(1, r2), band=plan.bands[0], check=False)   # ~10x the pairs
        pipe2.submit(1)
        skipped = pipe2.finish(1, check=True)
        skip_flags = int(pipe2.flags(1)[0].item())
        r2.render(stream, buf, gt_big, mt, cams[0], pipe2.begin(2, r2), band=plan.bands[0], check=False)   # grown buffers
        pipe2.submit(2)
        after = pipe2.finish(2, check=True)
        assert after is not None, "the frame after a skipped one must be rendered"
        after = after.clone()
        pipe2.drain()
        # frames in flight INSIDE the rank (bench.py --gpus N): two renderers on priority streams take the frames in
        # turn; begin / render / submit on the lane's stream, finish here on stream `s`; three gather buffers, the
        # forced RCCL collective ordered behind the lane, the next writer of a buffer behind its reader
        lanes = par.lanes(torch, gs, dev, 2)
        assert len(lanes) == 2 and lanes[0][1].native() != lanes[1][1].native()
        pipe3 = par.FramePipeline(torch, dist, plan, 0, W, "cuda", depth=3, force_collective=True)
        for b in pipe3.bufs:
            b.fill_(float("nan"))
        rounds = 3                                   # every buffer and every lane is reused several times
        got_lanes = []
        for j in range(rounds * frames):
            rr, gs_s, ts = lanes[j % 2]
            with torch.cuda.stream(ts):
                rr.render(gs_s, buf, gt, mt, cams[j % frames], pipe3.begin(j, rr), band=plan.bands[0], check=False)
                pipe3.submit(j)
            if j:
                prev = pipe3.finish(j - 1, check=True)
                assert prev is not None
                got_lanes.append(prev.clone())
        got_lanes.append(pipe3.finish(rounds * frames - 1, check=True).clone())
        pipe3.drain()
    s.synchronize()
    torch.cuda.synchronize()
    have_lanes = [hashlib.sha256(np.ascontiguousarray(t[:H].cpu().numpy()).tobytes()).hexdigest() for t in got_lanes]
    assert have_lanes == [want[j % frames] for j in range(rounds * frames)], "frames of the lanes differ: %s" % (
        [a == want[j % frames] for j, a in enumerate(have_lanes)],)
    for rr, gs_s, _ in lanes:
        rr.destroy()
        gs_s.close()
    assert skipped is None and skip_flags == 3, (skipped is None, skip_flags)
    r.render(own, buf, gt_big, mt, cams[0], img.device_ptr())
    want_big = hashlib.sha256(img.download(own, np.float32).tobytes()).hexdigest()
    assert hashlib.sha256(np.ascontiguousarray(after[:H].cpu().numpy()).tobytes()).hexdigest() == want_big
    r.wait_frame()
    have = [hashlib.sha256(np.ascontiguousarray(t[:H].cpu().numpy()).tobytes()).hexdigest() for t in got]
    assert have == want, "frames out of the RCCL pipeline differ from the plain frames: %s" % (
        [a == b for a, b in zip(have, want)],)
    assert hashlib.sha256(np.ascontiguousarray(sync_img[:H].cpu().numpy()).tobytes()).hexdigest() == want[0]
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps({"ok": True, "order": order, "frames": frames, "backend": "nccl", "world_size": 1,
                      "work_handle": sorted(kinds), "skipped_frame_dropped": True, "lanes_frames": len(have_lanes), "hip_runtime": hiprt.info()["source"],
                      "libamdhip64": mapped["libamdhip64"], "libhsa": mapped["libhsa-runtime64"],
                      "torch": torch.__version__, "hip": torch.version.hip}), flush=True)


if __name__ == "__main__":
    main()
