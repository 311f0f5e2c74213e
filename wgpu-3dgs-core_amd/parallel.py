"""Multi-GPU frame sharding: the scene is replicated, the IMAGE is
partitioned by 16-pixel tile rows, one process per GPU, and ONE all-gather of the f32 RGBA rows
(torch.distributed, backend "nccl" = RCCL over xGMI) hands every rank every band.

Row sharding changes no per-pixel arithmetic and no per-tile order, so the N-GPU image equals the
1-GPU image bit for bit.  PyTorch is used here only for device memory and the collective.

Bands are contiguous runs of tile rows.  The default plan gives rank g the rows
[floor(g R / G), floor((g + 1) R / G)) (8/9 rows at 1080p on 8 GPUs).  Tile rows are
not equally expensive — the rows through the image centre hold several times the pairs of the
border rows — so `BandPlan.rebalanced` re-cuts the bands from per-row costs measured on an
earlier frame (pairs per tile row) so that every rank gets about the same cost.

Exchange layout: every rank renders its band straight into ITS chunk of a gather buffer of
G equal chunks (chunk = the tallest band, in pixel rows, + ONE flag row); one in-place all-gather
fills the other chunks; `assemble` returns the contiguous image.  Equal chunks keep the exchange a
single fixed-size collective whatever the plan.

Skipped bands: a frame whose band needs more (tile, Gaussian) pairs than the renderer's buffers
hold is SKIPPED on the device (include/gs3d.h, GS_ERR_PAIR_CAPACITY: the band's rows keep what they
held).  On one GPU the viewer sees that in the frame result; here the other ranks must learn it too,
or the all-gather delivers a torn image (this rank's chunk from an older frame, the others' from
this one).  The renderer therefore writes each frame's flags word into the first word of the chunk's
flag row (`gs_renderer_set_frame_flags_target`, no extra launch), the ONE collective carries it to
every rank, and `frame_flags` / `FramePipeline.finish(i, check=True)` tell the caller to drop the
frame — on all ranks alike, since all of them see the same G words.

One HIP runtime: the torch wheel bundles its own libamdhip64 / libhsa-runtime64 and the product
library must run on the SAME copy as torch and RCCL (two runtimes in one process: the second finds
no GPU).  `_capi.load()` arranges that for either import order (`_hiprt.py`: torch's copy is mapped
first whenever torch is installed); every entry point here that receives the torch module re-checks
it (`_hiprt.check`) so that a process that forced GS3D_HIP_RUNTIME=system and then imported torch
fails with a clear message instead of "No HIP GPUs are available".

Streams: launch on a dedicated non-default stream (`torch.cuda.Stream()`, wrapped with
`Device.wrap_stream(s.cuda_stream)` and made current with `torch.cuda.stream(s)` around the calls of
this module): RCCL orders its own stream behind the CURRENT stream when a collective is issued and
`work.wait()` makes the CURRENT stream wait for it; the legacy null stream would serialise with every
other stream of the device and defeat the overlap `FramePipeline` exists for.  Give RCCL's stream a priority of
its own as well (`ProcessGroupNCCL.Options().is_high_priority_stream = True`, bench.py): HIP keeps separate hardware
queues per priority level, while two streams of ONE priority may share a queue — the all-gather would then sit in
front of the next frame's kernels instead of running beside them (measured on one GPU with two render streams:
no overlap at all until they had different priorities).
"""
import numpy as np

from . import _hiprt

TILE = 16


class BandPlan:
    """bands[g] = (ty0, ty1): tile rows of rank g; chunk_rows = pixel rows of one gather chunk."""

    def __init__(self, height, world_size, bands=None):
        self.height = int(height)
        self.world_size = int(world_size)
        self.tiles_y = (self.height + TILE - 1) // TILE
        if bands is None:
            r, g = self.tiles_y, self.world_size
            bands = [((k * r) // g, ((k + 1) * r) // g) for k in range(g)]
        self.bands = [(int(a), int(b)) for a, b in bands]
        assert len(self.bands) == self.world_size
        assert self.bands[0][0] == 0 and self.bands[-1][1] == self.tiles_y
        assert all(self.bands[i][1] == self.bands[i + 1][0] for i in range(self.world_size - 1))
        assert all(b >= a for a, b in self.bands)
        # rows of one gather chunk: the tallest band + one row whose first word carries the band's frame flags
        self.band_rows = max(1, max(b - a for a, b in self.bands)) * TILE
        self.chunk_rows = self.band_rows + 1

    def pixel_rows(self, rank):
        """[y0, y1) of rank's band in image rows (clipped to the image height)"""
        a, b = self.bands[rank]
        return a * TILE, min(b * TILE, self.height)

    def rebalanced(self, row_cost):
        """A plan whose bands carry about equal cost.  row_cost[ty] >= 0 is the measured cost of tile
        row ty (e.g. pairs in that row plus a constant for the per-row fixed work).  Greedy cut of the
        cumulative cost at k/G of the total; every rank keeps at least one row while rows remain."""
        cost = np.asarray(row_cost, dtype=np.float64)
        assert cost.shape == (self.tiles_y,) and (cost >= 0).all()
        g, r = self.world_size, self.tiles_y
        cum = np.concatenate([[0.0], np.cumsum(cost)])
        total = cum[-1]
        if total <= 0 or g == 1:
            return BandPlan(self.height, g)
        cuts = [0]
        for k in range(1, g):
            target = total * k / g
            c = int(np.searchsorted(cum, target, side="left"))
            if c > 0 and abs(cum[c - 1] - target) <= abs(cum[min(c, r)] - target):
                c -= 1                                   # the cut nearest to the target
            lo = cuts[-1] + (1 if r >= g else 0)         # every rank keeps a row while there are enough rows
            hi = r - (g - k) if r >= g else r
            cuts.append(min(max(c, lo), hi))
        cuts.append(r)
        return BandPlan(self.height, g, list(zip(cuts[:-1], cuts[1:])))


def band_plan(height, world_size):
    """(bands, padded pixel rows of the gather buffer) of the default plan"""
    p = BandPlan(height, world_size)
    return p.bands, p.chunk_rows * world_size


def allocate_gather(torch, plan, width, device):
    """Gather buffer [G * chunk_rows, width, 4] f32: chunk g holds band g from its first row on;
    the rows of a chunk past its band's height are never written."""
    _hiprt.check("torch")
    return torch.zeros((plan.world_size * plan.chunk_rows, width, 4), dtype=torch.float32, device=device)


def band_target_ptr(buf, plan, rank, width):
    """Device pointer to hand to the renderer as the FULL-frame base so that image row y of rank's
    band lands on row (y - band_y0) of rank's chunk: the renderer only ever writes the band's rows,
    so the (virtual) rows above them are never touched."""
    y0, _ = plan.pixel_rows(rank)
    return buf.data_ptr() + (rank * plan.chunk_rows - y0) * width * 16


def flags_ptr(buf, plan, rank, width):
    """Device pointer of the flags word of rank's chunk (first word of the chunk's last row): hand it
    to `Renderer.set_frame_flags_target` before rendering the band into this buffer."""
    return buf.data_ptr() + ((rank + 1) * plan.chunk_rows - 1) * width * 16


def frame_flags(torch, buf, plan):
    """[G] int32 device tensor: the flags word of every band of the (gathered) frame in `buf`;
    0 = rendered, bit 1 (FRAME_FLAG_SKIPPED) = that band was not written."""
    c = plan.chunk_rows
    return buf.view(torch.int32)[c - 1::c, 0, 0]


def gather_bands(dist, buf, plan, rank, async_op=False, force=False):
    """One all-gather, in place: every rank contributes its chunk of `buf`.  The form of the
    collective is chosen once from the backend (never by catching an error from a collective: ranks
    that disagree about which collective they are in deadlock).  async_op=True (RCCL only) returns
    the work handle: the collective runs on its own stream behind the render already enqueued, and
    the caller's stream goes on without waiting for it until handle.wait().  A world of one has
    nothing to exchange and returns at once unless force=True, which issues the collective anyway
    (tests: the real RCCL call, its in-place aliasing and its stream semantics on ONE GPU)."""
    g = plan.world_size
    if g == 1 and not force:
        return None
    c = plan.chunk_rows
    mine = buf[rank * c:(rank + 1) * c]
    if dist.get_backend() == "nccl":        # RCCL: fused form, output aliases the input chunk
        return dist.all_gather_into_tensor(buf, mine, async_op=async_op) if async_op else \
            dist.all_gather_into_tensor(buf, mine)
    # gloo (rehearsals only): list form; gloo has no all-gather on device tensors, so a device
    # buffer is staged through the host (this is NOT the product path: RCCL above is)
    on_device = buf.device.type != "cpu"
    src = mine.cpu() if on_device else mine.contiguous()
    recv = [src.new_empty(src.shape) for _ in range(g)]
    dist.all_gather(recv, src)
    for r in range(g):
        if r != rank:
            buf[r * c:(r + 1) * c].copy_(recv[r])


def assemble(torch, buf, plan):
    """The contiguous [height, width, 4] image from the gathered chunks (one device copy)."""
    c = plan.chunk_rows
    rows = []
    for r in range(plan.world_size):
        y0, y1 = plan.pixel_rows(r)
        if y1 > y0:
            rows.append(buf[r * c:r * c + (y1 - y0)])
    return rows[0] if len(rows) == 1 else torch.cat(rows, dim=0)


class FramePipeline:
    """Frames in flight on `depth` gather buffers used in turn: the all-gather of frame i runs on the
    collective's stream while frame i + 1 is rendered into the next buffer (xGMI transfer hidden
    behind compute; a frame costs max(render, gather) instead of their sum).

        buf_ptr = pipe.begin(i, renderer)   # base pointer for the renderer (waits for the frame that last used the
                                            # buffer; points the renderer's frame-flags word into this buffer's chunk)
        ... render frame i's band into buf_ptr ...
        pipe.submit(i)                 # start the exchange of frame i
        image = pipe.finish(i - 1)     # the complete previous frame (orders the caller's stream behind its exchange)

    begin / render / submit may run on another stream than finish (several renderers on priority streams taking the
    frames in turn, `lanes()` below: a rank's band is a chain of small dependent kernels, and three of them in flight
    cost 0.076 ms per frame instead of 0.162 at 1 M / 8 ranks): the exchange is ordered behind the stream that is
    current at submit, the buffer's next writer behind the stream that consumed it.

    A band that overflowed its pair capacity is skipped on the device; its flags word travels with the
    band.  `finish(i, check=True)` reads the G words back (one small device-to-host copy: a presenting
    viewer synchronises here anyway) and returns None when ANY rank skipped — every rank sees the same
    words, so all ranks drop the same frames and nobody presents a torn image; the next frame already has
    the grown buffers.  `finish(i)` (check=False) returns the image without looking; `flags(i)` returns
    the device tensor of the words for callers that accumulate them without a host round trip.
    """

    def __init__(self, torch, dist, plan, rank, width, device, depth=2, force_collective=False):
        assert depth >= 2
        self.torch, self.dist, self.plan, self.rank, self.width = torch, dist, plan, rank, width
        self.force = bool(force_collective)
        self.bufs = [allocate_gather(torch, plan, width, device) for _ in range(depth)]
        self.work = [None] * depth          # exchange in flight on each buffer
        self.frame = [None] * depth         # frame number each buffer holds
        self.reader = [None] * depth        # stream on which finish() / flags() last read each buffer
        self.gathered = [None] * depth      # event recorded by the first stream that waited for the buffer's exchange
        self.submitted = [None] * depth     # event behind everything submit() enqueued on ITS stream (the render, and
                                            # — gloo rehearsals — the staged copies of the other ranks' bands)

    def _slot(self, i):
        return i % len(self.bufs)

    def _wait(self, k):
        if self.work[k] is not None:
            self.work[k].wait()             # stream-level: the current stream waits, the host does not
            self.work[k] = None
            if self.bufs[k].device.type != "cpu":
                # the handle is gone after this: a LATER reader on another stream (flags(i) here, finish(i) there)
                # must still be ordered behind the collective, not just behind the render — it waits for this event
                self.gathered[k] = self.torch.cuda.Event()
                self.gathered[k].record(self.torch.cuda.current_stream())
        elif self.gathered[k] is not None:
            self.torch.cuda.current_stream().wait_event(self.gathered[k])     # (a no-op on the stream that recorded it)
        if self.submitted[k] is not None:
            self.torch.cuda.current_stream().wait_event(self.submitted[k])     # (a no-op on the stream that submitted)

    def _note_reader(self, k):
        # frames may be rendered on other streams than the one that consumes them (frames in flight on priority
        # streams): remember who read the buffer, so that its next writer can be ordered behind that stream
        if self.bufs[k].device.type != "cpu":
            self.reader[k] = self.torch.cuda.current_stream()

    def begin(self, i, renderer=None):
        k = self._slot(i)
        self._wait(k)                       # the buffer's previous exchange must have read and written it
        rd = self.reader[k]
        if rd is not None:
            # ... and whoever consumed that frame (assemble's copy, the flags OR) must be done reading: everything the
            # reader's stream has been given so far — the views finish() / flags() handed out are used right away
            cur = self.torch.cuda.current_stream()
            if rd != cur:
                ev = self.torch.cuda.Event()
                ev.record(rd)
                cur.wait_event(ev)
            self.reader[k] = None
        self.frame[k] = i
        if renderer is not None:
            renderer.set_frame_flags_target(flags_ptr(self.bufs[k], self.plan, self.rank, self.width))
        return band_target_ptr(self.bufs[k], self.plan, self.rank, self.width)

    def submit(self, i):
        k = self._slot(i)
        assert self.frame[k] == i
        self.work[k] = gather_bands(self.dist, self.bufs[k], self.plan, self.rank, async_op=True, force=self.force)
        self.submitted[k] = None
        self.gathered[k] = None
        if self.bufs[k].device.type != "cpu":
            self.submitted[k] = self.torch.cuda.Event()
            self.submitted[k].record(self.torch.cuda.current_stream())

    def flags(self, i):
        """device tensor [G] of frame i's per-band flags (valid once its exchange has been waited for)"""
        k = self._slot(i)
        assert self.frame[k] == i, "frame %d is no longer in the pipeline" % i
        self._wait(k)
        self._note_reader(k)
        return frame_flags(self.torch, self.bufs[k], self.plan)

    def finish(self, i, check=False):
        k = self._slot(i)
        assert self.frame[k] == i, "frame %d is no longer in the pipeline" % i
        self._wait(k)
        self._note_reader(k)
        if check and bool((frame_flags(self.torch, self.bufs[k], self.plan) != 0).any().item()):
            return None                     # some band was skipped: drop the frame on every rank
        return assemble(self.torch, self.bufs[k], self.plan)

    def drain(self):
        for k in range(len(self.bufs)):
            self._wait(k)


def lanes(torch, gs, device, count, first_renderer=None, priorities=None):
    """`count` (renderer, gs stream, torch stream) triples for frames in flight inside one rank: the streams are
    created with DIFFERENT priorities (gs_stream_create_with_priority: different hardware queues) and handed to torch
    as external streams, so that `with torch.cuda.stream(lane[2])` makes RCCL and FramePipeline order themselves behind
    the lane.  Default priorities: default and least — the greatest is left to RCCL's stream
    (ProcessGroupNCCL.Options.is_high_priority_stream)."""
    least, greatest = device.stream_priority_range()
    if priorities is None:
        priorities = [0, least] if least != greatest else [0]
    out = []
    for k in range(max(1, min(int(count), len(priorities)))):
        s = device.create_stream(priority=priorities[k])
        out.append((first_renderer if (k == 0 and first_renderer is not None) else gs.Renderer(device), s,
                    torch.cuda.ExternalStream(s.native())))
    return out


def render_sharded(dist, torch, buf, plan, rank, width, render_band, renderer=None, check=True):
    """render_band((ty0, ty1), full_frame_base_ptr) must render this rank's tile rows; then the
    bands are exchanged and the contiguous image is returned.  With `renderer` given its frame flags
    travel with the band, and (check=True) the result is None when any rank's band was skipped."""
    if renderer is not None:
        renderer.set_frame_flags_target(flags_ptr(buf, plan, rank, width))
    render_band(plan.bands[rank], band_target_ptr(buf, plan, rank, width))
    gather_bands(dist, buf, plan, rank)
    if renderer is not None and check and bool((frame_flags(torch, buf, plan) != 0).any().item()):
        return None
    return assemble(torch, buf, plan)


def row_costs_from_ranges(ranges, tiles_x, tiles_y, fixed=64.0):
    """Cost per tile row from a frame's per-tile [start, end) ranges (gs_renderer_download_ranges):
    pairs in the row plus a constant per tile for the fixed work."""
    r = np.asarray(ranges, dtype=np.int64).reshape(tiles_y, tiles_x, 2)
    return (r[..., 1] - r[..., 0]).sum(axis=1).astype(np.float64) + fixed * tiles_x
