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


# ---- balanced variant: interleaved row blocks -------------------------------------------------
def interleave_ok(world, height, block_rows):
    return world > 0 and block_rows > 0 and height % (block_rows * world) == 0


def interleaved_rows(rank, world, height, block_rows):
    """Image rows owned by `rank`, in the order they are stored in its compact tile."""
    assert interleave_ok(world, height, block_rows)
    rows = []
    for g in range(height // (block_rows * world)):
        y = (g * world + rank) * block_rows
        rows.extend(range(y, y + block_rows))
    return rows


class InterleavedGather:
    """One RCCL gather per frame of compact interleaved tiles, multi-buffered so that the gather of
    frame k overlaps the renders of the following frames (`tile()` hands out the buffer to render into after
    making the CURRENT stream wait for whatever last used it; `submit()` orders the gather after whatever the
    current stream has enqueued).  On rank 0 EVERY frame ends in its final layout: the gathered blocks are put in
    row order (one 4*w*h-byte device copy, on the stream the frame was rendered on) before the frame counts as
    done -- the film is not complete while its blocks are still rank-major.  Callers that alternate two render
    streams (bench.py) call both inside `with torch.cuda.stream(s)`."""

    def __init__(self, width, height, rank, world, block_rows, device, group=None, always_gather=False, buffers=3, all_ranks=False):
        assert interleave_ok(world, height, block_rows)
        self.w, self.h, self.rank, self.world, self.b, self.group = width, height, rank, world, block_rows, group
        self.collective = world > 1 or always_gather  # always_gather: run the collective even at world size 1 (rehearsal)
        # all_ranks: ONE all-gather instead of ONE gather -- every rank ends with the whole film in row order (SURVEY.md 8(e):
        # "equivalently ncclAllGather if every rank wants the image"); the default is the gather to rank 0
        self.all_ranks = bool(all_ranks)
        self.owner = self.all_ranks or rank == 0  # this rank assembles films
        self.rows = height // world
        # three buffer sets: the gather of frame k may take until frame k+3's render wants its buffers back
        self.nbuf = max(2, int(buffers))
        self.tiles = [torch.zeros((self.rows, width, 4), dtype=torch.uint8, device=device) for _ in range(self.nbuf)]
        self.pending = [None] * self.nbuf   # gather still reading tiles[i] (ranks other than 0)
        self.placed = [None] * self.nbuf    # rank 0, CUDA: event after frame i's blocks were put in row order
        self.k = 0
        self.recv = None
        self.out = None
        self.cuda = torch.device(device).type == "cuda"
        if self.owner and self.collective:
            self.recv = [torch.empty((world, self.rows, width, 4), dtype=torch.uint8, device=device) for _ in range(self.nbuf)]
            self.out = [torch.empty((height, width, 4), dtype=torch.uint8, device=device) for _ in range(self.nbuf)]

    def tile(self):
        i = self.k % self.nbuf
        if self.pending[i] is not None:
            self.pending[i].wait()  # stream-level wait: the buffer is free again
            self.pending[i] = None
        if self.placed[i] is not None:
            self.placed[i].wait()   # (the frame that used this set may have been rendered on the other stream)
            self.placed[i] = None
        return self.tiles[i]

    def submit(self):
        """Gather the tile handed out by the last tile() call; on rank 0 also put the frame in row order."""
        i = self.k % self.nbuf
        self.k += 1
        if not self.collective:
            if self.cuda:  # the buffer is handed out again after nbuf frames, possibly to another stream: order that use after this render
                self.placed[i] = torch.cuda.Event()
                self.placed[i].record()
            return
        if self.all_ranks:
            work = dist.all_gather_into_tensor(self.recv[i].view(-1), self.tiles[i].view(-1), group=self.group, async_op=True)
        else:
            glist = [self.recv[i][r] for r in range(self.world)] if self.rank == 0 else None
            work = dist.gather(self.tiles[i], gather_list=glist, dst=0, group=self.group, async_op=True)
        if not self.owner:
            self.pending[i] = work
            return
        work.wait()  # (stream-level with RCCL: the current stream continues after the gather; the other stream renders meanwhile)
        g = self.h // (self.b * self.world)
        self.out[i].view(g, self.world, self.b, self.w, 4).copy_(
            self.recv[i].view(self.world, g, self.b, self.w, 4).permute(1, 0, 2, 3, 4))
        if self.cuda:
            self.placed[i] = torch.cuda.Event()
            self.placed[i].record()

    def finish(self):
        """Wait for outstanding gathers; on rank 0 (every rank with all_ranks) return the last frame's (height, width, 4) film."""
        for i in range(self.nbuf):
            if self.pending[i] is not None:
                self.pending[i].wait()
                self.pending[i] = None
            if self.placed[i] is not None:
                self.placed[i].wait()
                self.placed[i] = None
        if not self.owner:
            return None
        last = (self.k - 1) % self.nbuf
        return self.out[last] if self.collective else self.tiles[last]
