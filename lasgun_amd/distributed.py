"""Row-tile sharding of one film across the GPUs of a node, one process per GPU.

The reference partitions pixels by stride over CPU threads (src/lib.rs:110-162); pixels are
independent, so any partition gives the same image.  Here rank r renders the contiguous row
tile `row_tile(r, world, h)` with the scene replicated on every GPU, and ONE collective at the
end gathers the RGBA8 tiles to rank 0 (RCCL over xGMI when the backend is "nccl": a fan-in over
distinct links, 4*w*h/world bytes per rank).  No other communication exists on this path.
"""
import torch
import torch.distributed as dist


def row_tile(rank, world, height):
    """Rows [y0, y1) of rank `rank`: contiguous, covering, sizes differ by at most one row."""
    base, rem = divmod(height, world)
    y0 = rank * base + min(rank, rem)
    y1 = y0 + base + (1 if rank < rem else 0)
    return y0, y1


def max_tile_rows(world, height):
    return (height + world - 1) // world


def gather_tiles(tile, width, height, rank, world, group=None):
    """Gather per-rank uint8 tiles (rows_r, width, 4) to rank 0 -> (height, width, 4) or None.

    Tiles are padded to the common maximum row count so a single fixed-size gather suffices.
    """
    if world == 1:
        return tile
    rows = max_tile_rows(world, height)
    y0, y1 = row_tile(rank, world, height)
    if tile.shape[0] == rows:
        send = tile.contiguous()
    else:
        send = torch.zeros((rows, width, 4), dtype=torch.uint8, device=tile.device)
        send[: y1 - y0] = tile
    recv = None
    if rank == 0:
        recv = [torch.empty((rows, width, 4), dtype=torch.uint8, device=tile.device) for _ in range(world)]
    try:
        dist.gather(send, gather_list=recv, dst=0, group=group)
    except (RuntimeError, NotImplementedError):  # backend without gather: fall back to all_gather
        allr = [torch.empty((rows, width, 4), dtype=torch.uint8, device=tile.device) for _ in range(world)]
        dist.all_gather(allr, send, group=group)
        recv = allr if rank == 0 else None
    if rank != 0:
        return None
    out = torch.empty((height, width, 4), dtype=torch.uint8, device=tile.device)
    for r in range(world):
        a, b = row_tile(r, world, height)
        out[a:b] = recv[r][: b - a]
    return out
