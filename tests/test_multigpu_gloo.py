"""world_size-2 (and 3) gloo test of the tile sharding + frame gather that bench.py uses across GPUs.

No GPU here: each rank renders ITS tiles with the CPU oracle (test infrastructure standing in for the
device) into a zero frame; after the FrameGather (pack, all-gather, scatter) every rank must hold the bits of
a single-process render, and the ray counters must add up. The dealer itself is checked for balance and for
not degenerating into column stripes at the bench's 120-tile rows.
"""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_lib as O
    rtd = importlib.import_module("ray-tracing_amd.dist")
    r, w = rtd.init("gloo")
    assert (r, w) == (rank, world)
    W, H = 80, 44                     # 44 rows: the last tile row is partial
    scene = O.Scene("reflexion", 3, 12)
    cam = O.default_camera(W / H)
    prm = O.make_params(W, H, spp=2, jitter=1, seed=4)
    frame = torch.zeros((H, W, 3), dtype=torch.float32)
    rays = 0
    for (x, y, tw, th) in rtd.shard_tiles(W, H, rank, world, 16):
        rgb, prim, st, _ = scene.render(cam, prm, x, y, tw, th, threads=1)
        frame[y:y + th, x:x + tw] = torch.from_numpy(rgb)
        rays += st.rays_primary + st.rays_secondary + st.rays_shadow
    gather = rtd.FrameGather(W, H, world, "cpu", 16)
    assert gather.bytes_sent_per_rank() < W * H * 12        # a slab, not the frame
    gather(frame, rank)
    total = rtd.sum_counters([rays])[0]
    slow = rtd.max_over_ranks(float(rank))
    np.save(os.path.join(out_dir, f"frame_{rank}.npy"), frame.numpy())
    with open(os.path.join(out_dir, f"rays_{rank}.txt"), "w") as f:
        f.write(f"{total} {slow}")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_sharded_render_gathers_to_the_single_process_frame(world, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    W, H = 80, 44
    scene = O.Scene("reflexion", 3, 12)
    full, _, st, _ = scene.render(O.default_camera(W / H), O.make_params(W, H, spp=2, jitter=1, seed=4), threads=4)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / f"frame_{r}.npy")
        assert np.array_equal(got.view(np.uint32), full.view(np.uint32)), f"rank {r}: gathered frame differs"
        total, slow = open(tmp_path / f"rays_{r}.txt").read().split()
        assert int(total) == st.rays_primary + st.rays_secondary + st.rays_shadow
        assert float(slow) == world - 1


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_tile_deal_is_a_balanced_partition_and_not_column_stripes(world):
    rtd = importlib.import_module("ray-tracing_amd.dist")
    W, H, T = 3840, 2160, 32                                  # the bench frame: 120 tiles per row, 68 rows (the last one partial)
    ntx, nty = (W + T - 1) // T, (H + T - 1) // T
    owner = np.full((nty, ntx), -1)
    for r in range(world):
        tiles = rtd.shard_tiles(W, H, r, world, T)
        for (x, y, w, h) in tiles:
            assert owner[y // T, x // T] == -1                # disjoint
            owner[y // T, x // T] = r
            assert w == min(T, W - x) and h == min(T, H - y)
    assert (owner >= 0).all()                                 # complete
    counts = np.bincount(owner.ravel(), minlength=world)
    assert counts.max() - counts.min() <= nty                 # at most one tile per row apart (120 % 8 == 0: exactly equal)
    # a one-tile-wide vertical feature must not land on one rank: every column holds every rank, none dominates
    for tx in range(ntx):
        col = np.bincount(owner[:, tx], minlength=world)
        assert (col > 0).all() and col.max() <= 0.5 * nty + 1, (tx, col)
    # same for a horizontal feature
    for ty in range(nty):
        row = np.bincount(owner[ty], minlength=world)
        assert row.max() - row.min() <= 1
