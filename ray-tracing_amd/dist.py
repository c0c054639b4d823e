"""Multi-GPU plumbing: one process per GPU, image-space tiles sharded across ranks, one collective.

The render loop shards naturally (pixels are independent): every rank holds the whole
scene and renders the 32x32 tiles dealt to it. Tiles are dealt row by row, round-robin with a
pseudo-random offset per tile row (cost per pixel is heavy-tailed, SURVEY 6: never bands, and with
120 tiles per row and 2/4/8 ranks a plain round-robin would be column stripes). The frame is assembled
with ONE all-gather of packed per-rank pixel slabs (RCCL over xGMI on GPUs, gloo in the CPU tests)
followed by a device-side scatter: 1/world of the frame leaves every rank instead of the whole frame
an all-reduce would move. Disjoint tiles, copies only: the gathered frame has the bits of a
single-GPU render.
"""
import os

TILE = 32


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend, local_rank=None):
    """Initialises torch.distributed from the torchrun environment (no-op for one rank)."""
    import torch
    import torch.distributed as dist
    rank, world = rank_world()
    if world > 1 and not dist.is_initialized():
        kw = {}
        if backend == "nccl" and local_rank is not None:
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world


def row_offset(ty, world):
    """Offset of tile row ty in the round-robin deal (a fixed integer hash, the same on every rank)."""
    h = (ty * 0x9E3779B1 + 0x7F4A7C15) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h * 0x2C1B3C6D) & 0xFFFFFFFF
    h ^= h >> 12
    return h % world


def tile_owner(tx, ty, world):
    return (tx + row_offset(ty, world)) % world


def shard_tiles(width, height, rank, world, tile=TILE):
    """The tiles (x, y, w, h) of `rank`, in row-major order."""
    out = []
    for ty, y in enumerate(range(0, height, tile)):
        for tx, x in enumerate(range(0, width, tile)):
            if tile_owner(tx, ty, world) == rank:
                out.append((x, y, min(tile, width - x), min(tile, height - y)))
    return out


class FrameGather:
    """Assembles the full frame from the ranks' disjoint tile sets: pack -> all-gather -> scatter.

    Built once per (image size, world): per rank the flat pixel indices of its tiles (tile after tile),
    padded to the longest rank by repeating its first pixel (a repeated index writes the same value
    twice). `frame` is this rank's (H, W, 3) float32 tensor whose own tiles are rendered; after
    __call__ it holds the whole image on every rank."""

    def __init__(self, width, height, world, device, tile=TILE):
        import torch
        self.world = world
        self.shape = (height, width, 3)
        per_rank = []
        for r in range(world):
            idx = []
            for (x, y, w, h) in shard_tiles(width, height, r, world, tile):
                rows = torch.arange(y, y + h, dtype=torch.int64).unsqueeze(1) * width
                cols = torch.arange(x, x + w, dtype=torch.int64).unsqueeze(0)
                idx.append((rows + cols).reshape(-1))
            per_rank.append(torch.cat(idx) if idx else torch.zeros(0, dtype=torch.int64))
        self.counts = [int(t.numel()) for t in per_rank]
        assert sum(self.counts) == width * height
        self.slab = max(max(self.counts), 1)
        padded = []
        for t in per_rank:
            fill = t[:1] if t.numel() else torch.zeros(1, dtype=torch.int64)
            padded.append(torch.cat([t, fill.expand(self.slab - t.numel())]))
        self.index = torch.stack(padded).to(device)            # (world, slab)
        self.recv = torch.empty((world, self.slab, 3), dtype=torch.float32, device=device)

    def bytes_sent_per_rank(self):
        return self.slab * 12

    def __call__(self, frame, rank):
        import torch.distributed as dist
        if self.world == 1 or not (dist.is_available() and dist.is_initialized()):
            return frame
        assert tuple(frame.shape) == self.shape and frame.is_contiguous()
        flat = frame.view(-1, 3)
        mine = flat.index_select(0, self.index[rank])
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(self.recv.view(-1, 3), mine)
        else:
            parts = [self.recv[r] for r in range(self.world)]
            dist.all_gather(parts, mine)
        flat.index_copy_(0, self.index.view(-1), self.recv.view(-1, 3))
        return frame


def sample_range(rank, world, spp):
    """Sample sharding: the samples (first, count) of every pixel that `rank` renders; the ranges of all ranks tile 0 .. spp - 1."""
    first = rank * spp // world
    return first, (rank + 1) * spp // world - first


class SampleReduce:
    """Sample sharding: every rank renders a sample subset of EVERY pixel (rt_begin_frame_samples), the ranks'
    fixed-point accumulators are added with one all-reduce(sum) and their non-finite flag words OR-ed, and every rank resolves
    the same frame -- the bits of a single-GPU render, because the sums are integers. For frames in which a few pixels own most
    of the rays (the default scene: depth-50 recursion in the glass gorilla), where dealing tiles leaves ranks unequal work.

    RCCL has no bitwise OR: the nine flag bits of a pixel (NaN / +inf / -inf seen, per channel) travel as nine 7-bit counters in
    one int64 (world <= 64) and are turned back into bits after the sum."""

    def __init__(self, width, height, world, device):
        import torch
        assert world <= 64
        self.world = world
        self.acc = torch.zeros((height, width, 3), dtype=torch.int64, device=device)
        self.flags = torch.zeros((height, width), dtype=torch.int32, device=device)

    def bytes_reduced_per_rank(self):
        return self.acc.numel() * 8 + self.flags.numel() * 8

    @staticmethod
    def spread_flags(flags):
        """uint32 flag words (as int32) -> int64 with bit b of the word at bit 7 b."""
        import torch
        f = flags.to(torch.int64) & 0x1FF
        out = torch.zeros_like(f)
        for b in range(9):
            out |= ((f >> b) & 1) << (7 * b)
        return out

    @staticmethod
    def gather_flags(counts):
        """the inverse after a sum: bit b set iff counter b is not zero."""
        import torch
        out = torch.zeros_like(counts)
        for b in range(9):
            out |= (((counts >> (7 * b)) & 0x7F) != 0).to(torch.int64) << b
        return out.to(torch.int32)

    def reduce_tensors(self, acc, flags):
        """In place: acc (int64) summed over the ranks, flags (int32 flag words) OR-ed over the ranks."""
        import torch.distributed as dist
        if self.world == 1 or not (dist.is_available() and dist.is_initialized()):
            return acc, flags
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
        counts = self.spread_flags(flags)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        flags.copy_(self.gather_flags(counts))
        return acc, flags

    def __call__(self, ctx):
        """ctx: this rank's Context with its share of the samples queued or rendered. Afterwards ctx holds the whole frame's sums."""
        ctx.export_accumulators(self.acc.data_ptr(), self.flags.data_ptr())
        self.reduce_tensors(self.acc, self.flags)
        ctx.import_accumulators(self.acc.data_ptr(), self.flags.data_ptr())


def sum_counters(values, device=None):
    """All-reduce a small list of integer counters (ray counts)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return [int(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


def max_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
