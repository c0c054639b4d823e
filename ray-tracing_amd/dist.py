"""Multi-GPU plumbing: one process per GPU, image-space tiles sharded across ranks, one collective.

The render loop shards naturally (pixels are independent): every rank holds the whole
scene, renders an interleaved subset of the 32x32 tiles into a zero-initialised full-size frame and
the frames are summed with ONE all-reduce (RCCL over xGMI on GPUs, gloo in the CPU tests). Tiles are
disjoint and every other pixel is +0.0, so the sum is a bit-exact gather.
"""
import os


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


def shard_tiles(width, height, rank, world, tile=32):
    """Round-robin (interleaved) assignment: the cost per pixel is heavy-tailed, never use bands."""
    out = []
    k = 0
    for y in range(0, height, tile):
        for x in range(0, width, tile):
            if k % world == rank:
                out.append((x, y, min(tile, width - x), min(tile, height - y)))
            k += 1
    return out


def gather_frame(frame):
    """Sum of the per-rank partial frames (disjoint tiles) -> the full frame on every rank."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(frame, op=dist.ReduceOp.SUM)
    return frame


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
