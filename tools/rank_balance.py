#!/usr/bin/env python3
"""Strong-scaling rehearsal on ONE GPU: renders the tile set of every rank of an N-rank job in turn (same scene, same
frame) and prints the ranks' times -- the job's step time is the slowest rank's, so max/mean is the load imbalance the
tile deal leaves (ray-tracing_amd/dist.py).   python tools/rank_balance.py [--world 8] [--ntris N --width W --height H --spp S]"""
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
    a = ap.parse_args()
    rt = importlib.import_module("ray-tracing_amd")
    rtd = importlib.import_module("ray-tracing_amd.dist")
    hs = rt.HostScene(f"soup:{a.ntris}", 3, a.mesh_depth)
    ctx = rt.Context(0)
    ctx.upload_scene(hs)
    cam = rt.default_camera(a.width / a.height)
    prm = rt.make_params(a.width, a.height, spp=a.spp, max_depth=a.depth, jitter=1, seed=1)
    out = []
    for rank in [0] + list(range(a.world)):               # (rank 0 once more first: warm-up)
        tiles = rtd.shard_tiles(a.width, a.height, rank, a.world)
        ctx.set_camera(cam)
        ctx.begin_frame(prm)
        t0 = time.perf_counter()
        for t in tiles:
            ctx.render_tile(*t)
        ctx.flush()
        dt = time.perf_counter() - t0
        st = ctx.stats()
        out.append({"rank": rank, "tiles": len(tiles), "ms": round(1e3 * dt, 2), "rays": int(st.rays_primary + st.rays_secondary + st.rays_shadow)})
    out = out[1:]
    ms = [o["ms"] for o in out]
    print(json.dumps({"world": a.world, "ranks": out, "max_ms": max(ms), "mean_ms": round(sum(ms) / len(ms), 2),
                      "imbalance_max_over_mean": round(max(ms) / (sum(ms) / len(ms)), 4)}))
    ctx.close()


if __name__ == "__main__":
    main()
