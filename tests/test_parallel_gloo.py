"""world_size-2 (and 3) `gloo` test of the multi-GPU frame sharding logic on CPU:
band plan, per-rank band rendering, all-gather reassembly; the gathered frame must equal the
single-rank frame bit for bit.  The band renderer here is the CPU oracle (tests may use it); on
GPUs bench.py plugs the HIP renderer into the same functions."""
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


def _worker(rank, world, port, height, width, out_dir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
    import torch
    import torch.distributed as dist
    from importlib import import_module
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import synth
    from oracle import binding as ob
    par = import_module("wgpu_3dgs_core_amd.parallel")
    g = synth.scene(4000)
    pods = ob.pack(3, 0, g)
    cam = ob.camera_look_at((0, 0, 0), (0, 0, -1), (0, 1, 0), float(np.deg2rad(60)), width, height)
    gt, mt = ob.gaussian_transform(sh_deg=0), ob.model_transform()
    frame = par.allocate_frame(torch, height, width, world, "cpu")

    def render_band(band, fr):
        part = ob.render(3, 0, pods, gt, mt, cam, band=band)[0]
        y0, y1 = band[0] * 16, min(band[1] * 16, height)
        fr[y0:y1] = torch.from_numpy(part[y0:y1])

    img = par.render_sharded(dist, frame, rank, world, height, render_band)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), img.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_band_plan():
    from importlib import import_module
    sys.path[:0] = [ROOT]
    par = import_module("wgpu_3dgs_core_amd.parallel")
    rows, bands, padded = par.band_plan(1080, 8)
    assert rows == 9 and bands[0] == (0, 9) and bands[-1] == (63, 68) and padded == 1152
    assert par.band_plan(2160, 8)[1][-1] == (119, 135)
    for h in (16, 200, 1080, 2160):
        for w in (1, 2, 3, 4, 8):
            _, b, pad = par.band_plan(h, w)
            tiles_y = (h + 15) // 16
            assert b[0][0] == 0 and b[-1][1] == tiles_y and pad >= h
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_frame_equals_single_rank(tmp_path, world):
    import torch.multiprocessing as mp
    from oracle import binding as ob
    import synth
    height, width = 200, 320
    port = _free_port()
    mp.spawn(_worker, args=(world, port, height, width, str(tmp_path)), nprocs=world, join=True)
    g = synth.scene(4000)
    cam = ob.camera_look_at((0, 0, 0), (0, 0, -1), (0, 1, 0), float(np.deg2rad(60)), width, height)
    full = ob.render(3, 0, ob.pack(3, 0, g), ob.gaussian_transform(sh_deg=0), ob.model_transform(), cam)[0]
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert got.shape == full.shape
        assert np.array_equal(got.view(np.uint32), full.view(np.uint32)), "rank %d" % r
