#!/usr/bin/env python3
"""Strong-scaling rehearsal on ONE GPU: renders the tile set of every rank of an N-rank job in turn (same scene, same
frame) and prints the ranks' times -- the job's step time is the slowest rank's, so max/mean is the load imbalance the
tile deal leaves (ray-tracing_amd/dist.py).
    python tools/rank_balance.py [--world 8] [--ntris N --width W --height H --spp S]          (the C5 soup)
    python tools/rank_balance.py --scene final --width 480 --height 480 --spp 30 --depth 50 --mesh-depth 12 --world 8   (any named scene)"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--ntris", type=int, default=10_000_000)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--mesh-depth", type=int, default=24)
    ap.add_argument("--scene", default="", help="a named scene of the host mirror instead of the soup")
    ap.add_argument("--reps", type=int, default=1, help="frames per rank (the time quoted is their mean)")
    ap.add_argument("--shard", default="tiles", choices=["tiles", "samples"], help="tiles: every rank renders all samples of its tiles (dist.shard_tiles); samples: every rank renders its samples of all tiles (dist.sample_range, rt_begin_frame_samples)")
    a = ap.parse_args()
    rt = importlib.import_module("ray-tracing_amd")
    rtd = importlib.import_module("ray-tracing_amd.dist")
    hs = rt.HostScene(a.scene or f"soup:{a.ntris}", 3, a.mesh_depth)
    ctx = rt.Context(0)
    ctx.upload_scene(hs)
    cam = rt.default_camera(a.width / a.height)
    prm = rt.make_params(a.width, a.height, spp=a.spp, max_depth=a.depth, jitter=1, seed=1)
    out = []
    for rank in list(range(a.world)) + list(range(a.world)):     # (every rank once untimed first: buffers grown to the largest share, kernels loaded)
        tiles = rtd.shard_tiles(a.width, a.height, rank, a.world) if a.shard == "tiles" else rtd.shard_tiles(a.width, a.height, 0, 1)
        ctx.set_camera(cam)
        dt = 0.0
        for _ in range(a.reps):
            if a.shard == "tiles":
                ctx.begin_frame(prm)
            else:
                ctx.begin_frame_samples(prm, *rtd.sample_range(rank, a.world, a.spp))
            t0 = time.perf_counter()
            ctx.render_tiles(tiles)
            ctx.flush()
            dt += (time.perf_counter() - t0) / a.reps
        st = ctx.stats()
        out.append({"rank": rank, "tiles": len(tiles), "ms": round(1e3 * dt, 2), "rays": int(st.rays_primary + st.rays_secondary + st.rays_shadow)})
    out = out[a.world:]
    ms = [o["ms"] for o in out]
    print(json.dumps({"scene": a.scene or f"soup:{a.ntris}", "frame": f"{a.width}x{a.height}x{a.spp} depth {a.depth}", "shard": a.shard, "world": a.world, "ranks": out, "max_ms": max(ms), "mean_ms": round(sum(ms) / len(ms), 2),
                      "imbalance_max_over_mean": round(max(ms) / (sum(ms) / len(ms)), 4)}))
    ctx.close()


if __name__ == "__main__":
    main()
