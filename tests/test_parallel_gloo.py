"""world_size-2 (and 3) `gloo` test of the multi-GPU frame sharding logic on CPU:
band plans (default and cost-balanced), per-rank band rendering into the gather buffer, the
all-gather and the reassembly; the gathered frame must equal the single-rank frame bit for bit.
The band renderer here is the CPU oracle (tests may use it); on GPUs bench.py plugs the HIP
renderer into the same functions (tests/test_gpu_fullsize.py::test_sharded_path_single_gpu_4k drives
the HIP renderer through them on one GPU, and tests/test_gpu_rccl.py runs the real RCCL collective of
a world of one through FramePipeline)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _par():
    from importlib import import_module
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    return import_module("wgpu_3dgs_core_amd.parallel")


def _worker(rank, world, port, height, width, out_dir, balanced):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import synth
    from oracle import binding as ob
    par = _par()
    g = synth.scene(4000)
    pods = ob.pack(3, 0, g)
    cam = ob.camera_look_at((0, 0, 0), (0, 0, -1), (0, 1, 0), float(np.deg2rad(60)), width, height)
    gt, mt = ob.gaussian_transform(sh_deg=0), ob.model_transform()
    plan = par.BandPlan(height, world)
    if balanced:
        # per-row cost measured by every rank on its own band, exchanged, then a common re-cut
        proj, tiles = ob.preprocess(3, 0, pods, gt, mt, cam, band=plan.bands[rank])
        keys, _ = ob.build_keys(proj, tiles, (width + 15) // 16)
        rows = ((keys >> np.uint64(32)).astype(np.int64) // ((width + 15) // 16))
        mine = torch.from_numpy(np.bincount(rows, minlength=plan.tiles_y).astype(np.float64))
        dist.all_reduce(mine)
        plan = plan.rebalanced(mine.numpy() + 8.0)
    buf = par.allocate_gather(torch, plan, width, "cpu")
    flat = buf.view(-1)

    def render_band(band, base_ptr):
        part = ob.render(3, 0, pods, gt, mt, cam, band=band)[0]
        y0, y1 = band[0] * 16, min(band[1] * 16, height)
        # what the HIP renderer does with the base pointer: image row y -> base + y * width * 16 bytes
        off = (base_ptr - buf.data_ptr()) // 4
        for y in range(y0, y1):
            flat[off + y * width * 4:off + (y + 1) * width * 4] = torch.from_numpy(part[y].reshape(-1))

    img = par.render_sharded(dist, torch, buf, plan, rank, width, render_band)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), img.numpy())
    np.save(os.path.join(out_dir, "bands%d.npy" % rank), np.array(plan.bands))
    dist.barrier()
    dist.destroy_process_group()


def test_band_plan():
    par = _par()
    p = par.BandPlan(1080, 8)
    assert p.bands == [(0, 8), (8, 17), (17, 25), (25, 34), (34, 42), (42, 51), (51, 59), (59, 68)]   # floor(g R / G)
    assert p.band_rows == 9 * 16 and p.chunk_rows == 9 * 16 + 1      # tallest band + the flag row
    assert [b - a for a, b in par.BandPlan(2160, 8).bands] == [16, 17, 17, 17, 17, 17, 17, 17]
    for h in (16, 200, 1080, 2160):
        for w in (1, 2, 3, 4, 8):
            q = par.BandPlan(h, w)
            tiles_y = (h + 15) // 16
            assert q.bands[0][0] == 0 and q.bands[-1][1] == tiles_y
            assert all(q.bands[i][1] == q.bands[i + 1][0] for i in range(w - 1))
            assert q.chunk_rows * w >= min(h, tiles_y * 16)
            assert q.pixel_rows(w - 1)[1] == h


def test_rebalanced_plan_equalises_cost():
    par = _par()
    p = par.BandPlan(1080, 8)
    ty = np.arange(68)
    cost = 100.0 + 2000.0 * np.exp(-((ty - 33.5) / 12.0) ** 2)      # heavy centre rows
    q = p.rebalanced(cost)
    per = lambda plan: np.array([cost[a:b].sum() for a, b in plan.bands])
    assert per(q).max() < 0.75 * per(p).max()
    assert per(q).max() <= 1.25 * cost.sum() / 8
    assert all(b > a for a, b in q.bands)
    # degenerate inputs: zero cost -> default plan; fewer rows than ranks -> empty bands allowed
    assert p.rebalanced(np.zeros(68)).bands == p.bands
    small = par.BandPlan(32, 4).rebalanced(np.array([5.0, 1.0]))
    assert small.bands[0][0] == 0 and small.bands[-1][1] == 2
    # a single heavy row cannot be split: every other rank still gets work
    spike = np.ones(68); spike[10] = 1e6
    s = p.rebalanced(spike)
    assert all(b > a for a, b in s.bands)


@pytest.mark.parametrize("world,balanced", [(2, False), (3, False), (3, True)])
def test_sharded_frame_equals_single_rank(tmp_path, world, balanced):
    import torch.multiprocessing as mp
    from oracle import binding as ob
    import synth
    height, width = 200, 320
    port = _free_port()
    mp.spawn(_worker, args=(world, port, height, width, str(tmp_path), balanced), nprocs=world, join=True)
    g = synth.scene(4000)
    cam = ob.camera_look_at((0, 0, 0), (0, 0, -1), (0, 1, 0), float(np.deg2rad(60)), width, height)
    full = ob.render(3, 0, ob.pack(3, 0, g), ob.gaussian_transform(sh_deg=0), ob.model_transform(), cam)[0]
    bands0 = np.load(os.path.join(str(tmp_path), "bands0.npy"))
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert got.shape == full.shape
        assert np.array_equal(got.view(np.uint32), full.view(np.uint32)), "rank %d" % r
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "bands%d.npy" % r)), bands0)   # same plan everywhere


def _pipeline_worker(rank, world, port, height, width, out_dir, balanced=False):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = _par()
    plan = par.BandPlan(height, world)
    if balanced:
        # every rank re-cuts the same plan from the same (exchanged) per-row cost: heavy centre rows
        ty = np.arange(plan.tiles_y)
        cost = torch.from_numpy(100.0 + 2000.0 * np.exp(-((ty - 0.5 * plan.tiles_y) / (0.18 * plan.tiles_y)) ** 2)) / world
        dist.all_reduce(cost)
        plan = plan.rebalanced(cost.numpy())
        np.save(os.path.join(out_dir, "pbands%d.npy" % rank), np.array(plan.bands))
    pipe = par.FramePipeline(torch, dist, plan, rank, width, "cpu")
    y0, y1 = plan.pixel_rows(rank)
    images = []
    frames = 5

    class _Renderer:
        """stands in for gs.Renderer.set_frame_flags_target: remembers where the flags word goes"""
        target = None

        def set_frame_flags_target(self, ptr):
            self.target = ptr

    rr = _Renderer()
    dropped = []
    for i in range(frames):
        base = pipe.begin(i, rr)
        buf = pipe.bufs[i % 2]
        off = (base - buf.data_ptr()) // 4
        flat = buf.view(-1)
        # frame 3: the LAST rank's band overflows its pair capacity and is skipped — the rows keep what the
        # buffer held (frame 1) and the flags word says so
        skip = i == 3 and rank == world - 1
        if not skip:
            for y in range(y0, y1):          # frame i, row y: value 1000 i + y in every channel
                flat[off + y * width * 4:off + (y + 1) * width * 4] = float(1000 * i + y)
        assert rr.target == par.flags_ptr(buf, plan, rank, width)
        buf.view(torch.int32).view(-1)[(rr.target - buf.data_ptr()) // 4] = 3 if skip else 0
        pipe.submit(i)
        if i:
            img = pipe.finish(i - 1, check=True)
            dropped.append(img is None)
            images.append(torch.zeros(height, width, 4) if img is None else img.clone())
    img = pipe.finish(frames - 1, check=True)
    dropped.append(img is None)
    images.append(img.clone())
    pipe.drain()
    assert dropped == [False, False, False, True, False], dropped      # every rank drops frame 3, and only it
    np.save(os.path.join(out_dir, "pipe%d.npy" % rank), torch.stack(images).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_frame_pipeline_keeps_frames_apart(tmp_path, world):
    """FramePipeline: frame i + 1 is written into the other buffer while frame i is exchanged; every
    finished frame must hold exactly its own rows from every rank."""
    import torch.multiprocessing as mp
    height, width = 100, 7
    port = _free_port()
    mp.spawn(_pipeline_worker, args=(world, port, height, width, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), "pipe%d.npy" % rank))
        assert got.shape == (5, height, width, 4)
        for i in range(5):
            want = (1000 * i + np.arange(height, dtype=np.float32))[:, None, None] * np.ones((1, width, 4), np.float32)
            if i == 3:
                want[:] = 0.0          # dropped on every rank: one band was skipped (its flags word travelled with it)
            assert np.array_equal(got[i], want), (rank, i)


@pytest.mark.parametrize("balanced", [False, True])
def test_eight_ranks_on_the_4k_geometry(tmp_path, balanced):
    """First-contact insurance for the 8-GPU node: BASELINE config 4's geometry — 2160 rows = 135 tile
    rows over 8 ranks, bands of 16 / 17 tile rows (or re-cut to equal cost) — through FramePipeline with gloo: five
    pipelined frames, one band skipped in frame 3 and that frame dropped on all 8 ranks."""
    import torch.multiprocessing as mp
    par = _par()
    world, height, width = 8, 2160, 5
    base = par.BandPlan(height, world)
    assert sorted(b - a for a, b in base.bands) == [16] + [17] * 7 and base.bands[-1][1] == 135
    port = _free_port()
    mp.spawn(_pipeline_worker, args=(world, port, height, width, str(tmp_path), balanced), nprocs=world, join=True)
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), "pipe%d.npy" % rank))
        assert got.shape == (5, height, width, 4)
        for i in range(5):
            want = (1000 * i + np.arange(height, dtype=np.float32))[:, None, None] * np.ones((1, width, 4), np.float32)
            if i == 3:
                want[:] = 0.0
            assert np.array_equal(got[i], want), (rank, i)
    if balanced:
        bands = [np.load(os.path.join(str(tmp_path), "pbands%d.npy" % r)) for r in range(world)]
        assert all(np.array_equal(b, bands[0]) for b in bands)                    # one plan on every rank
        rows = bands[0][:, 1] - bands[0][:, 0]
        assert bands[0][0, 0] == 0 and bands[0][-1, 1] == 135 and (rows > 0).all()
        assert rows[3] < 17 and rows[0] > 17, rows                                # thin bands where the rows are heavy
