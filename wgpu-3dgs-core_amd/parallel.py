"""Multi-GPU frame sharding: the scene is replicated, the IMAGE is
partitioned by 16-pixel tile rows, one process per GPU, and one RCCL all-gather of the f32 RGBA
rows (torch.distributed, backend "nccl" = RCCL over xGMI) reassembles the frame on every rank.

Row sharding changes no per-pixel arithmetic and no per-tile order, so the N-GPU image equals the
1-GPU image bit for bit.  PyTorch is used here only for device memory and the collective.
"""
import numpy as np

TILE = 16


def band_plan(height, world_size):
    """Uniform bands of `rows` tile rows per rank (the last ranks may own fewer real rows).

    Returns (rows_per_rank, [(ty0, ty1) per rank], padded_height_px).  Uniform chunks make the
    exchange a single equal-sized all-gather with no per-rank size negotiation."""
    tiles_y = (height + TILE - 1) // TILE
    rows = (tiles_y + world_size - 1) // world_size
    bands = []
    for r in range(world_size):
        ty0 = min(r * rows, tiles_y)
        ty1 = min((r + 1) * rows, tiles_y)
        bands.append((ty0, ty1))
    return rows, bands, rows * world_size * TILE


def allocate_frame(torch, height, width, world_size, device):
    """Padded frame tensor [padded_height, width, 4] f32; rows >= height are never written."""
    _, _, padded = band_plan(height, world_size)
    return torch.zeros((padded, width, 4), dtype=torch.float32, device=device)


def gather_frame(dist, frame, rank, world_size, height):
    """All-gather every rank's band into `frame` in place (each rank contributes rows
    [rank*chunk, (rank+1)*chunk) of the padded frame)."""
    if world_size == 1:
        return
    rows, _, padded = band_plan(height, world_size)
    chunk = rows * TILE
    assert frame.shape[0] == padded
    mine = frame[rank * chunk:(rank + 1) * chunk]
    try:
        dist.all_gather_into_tensor(frame, mine)
    except (RuntimeError, NotImplementedError, AttributeError):
        # backends without the fused form (older gloo): list form on chunk views
        parts = [frame[r * chunk:(r + 1) * chunk] for r in range(world_size)]
        recv = [p if r != rank else p.clone() for r, p in enumerate(parts)]
        dist.all_gather(recv, mine.contiguous())
        for r, p in enumerate(parts):
            if r != rank:
                p.copy_(recv[r])


def render_sharded(dist, frame, rank, world_size, height, render_band):
    """render_band((ty0, ty1), frame) must render this rank's tile rows into `frame` (the padded
    full-frame tensor); then the bands are exchanged."""
    _, bands, _ = band_plan(height, world_size)
    render_band(bands[rank], frame)
    gather_frame(dist, frame, rank, world_size, height)
    return frame[:height]
