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


def _sample_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    rtd = importlib.import_module("ray-tracing_amd.dist")
    rtd.init("gloo")
    W, H = 40, 24
    g = torch.Generator().manual_seed(100 + rank)
    acc = torch.randint(-2**40, 2**40, (H, W, 3), generator=g, dtype=torch.int64)
    flags = torch.zeros((H, W), dtype=torch.int32)
    pick = torch.rand((H, W), generator=g) < 0.05
    flags[pick] = torch.randint(0, 512, (int(pick.sum()),), generator=g, dtype=torch.int32)
    np.save(os.path.join(out_dir, f"acc_in_{rank}.npy"), acc.numpy())
    np.save(os.path.join(out_dir, f"flags_in_{rank}.npy"), flags.numpy())
    red = rtd.SampleReduce(W, H, world, "cpu")
    red.reduce_tensors(acc, flags)
    np.save(os.path.join(out_dir, f"acc_out_{rank}.npy"), acc.numpy())
    np.save(os.path.join(out_dir, f"flags_out_{rank}.npy"), flags.numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sample_sharded_accumulators_reduce_to_sum_and_or(world, tmp_path):
    """The collective of the sample-sharded route (dist.SampleReduce): int64 sums added, flag words OR-ed (as 7-bit counters
    through a sum: RCCL has no bitwise OR). The render side of it -- two sample ranges add up to the bits of one frame -- needs
    the device and is tests/test_gpu_parity.py::test_sample_sharded_contexts_add_up_to_the_one_context_frame."""
    rtd = importlib.import_module("ray-tracing_amd.dist")
    assert [rtd.sample_range(r, 8, 30) for r in range(8)] == [(0, 3), (3, 4), (7, 4), (11, 4), (15, 3), (18, 4), (22, 4), (26, 4)]
    for spp in (1, 5, 16, 30):
        for w in (1, 2, 3, 8):
            rs = [rtd.sample_range(r, w, spp) for r in range(w)]
            assert rs[0][0] == 0 and sum(c for _, c in rs) == spp and all(rs[i][0] + rs[i][1] == rs[i + 1][0] for i in range(w - 1))
    f = torch.arange(512, dtype=torch.int32)
    assert torch.equal(rtd.SampleReduce.gather_flags(rtd.SampleReduce.spread_flags(f) * 64), f)       # 64 ranks with the same bit: no carry
    mp.spawn(_sample_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want_acc = sum(np.load(tmp_path / f"acc_in_{r}.npy") for r in range(world))
    want_flags = np.zeros_like(np.load(tmp_path / "flags_in_0.npy"))
    for r in range(world):
        want_flags |= np.load(tmp_path / f"flags_in_{r}.npy")
    assert want_flags.any()
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"acc_out_{r}.npy"), want_acc)
        assert np.array_equal(np.load(tmp_path / f"flags_out_{r}.npy"), want_flags)
