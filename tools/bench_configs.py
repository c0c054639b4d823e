#!/usr/bin/env python3
"""Side measurements on the BASELINE configs that bench.py does not quote.

    python tools/bench_configs.py C1|C2|C3|C4|C4cow|chamber [--steps K] [--spp N] [--depth D] [--ppm out.ppm]

One JSON line: ms/frame, Mrays/s split by ray class, time inside the two traversal kernels.
bench.py stays the C5 line the driver reads; the numbers printed here go into DESIGN.md.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (scene, top depth, mesh depth, width, height, spp, max_depth)
    "C1": ("final", 3, 12, 480, 480, 30, 50),
    "C2": ("final", 3, 12, 480, 480, 16, 50),
    "C3": ("refraction", 3, 12, 1920, 1080, 30, 5),
    "C4": ("mesh_gorilla", 3, 12, 1920, 1080, 30, 50),
    "C4cow": ("mesh_cow", 3, 12, 1920, 1080, 30, 50),
    # the normal-mapped mirror chamber (Scene.h:601-714) at the reference's window size and literals
    "chamber": ("reflexion_chamber", 3, 12, 480, 480, 30, 50),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=sorted(CONFIGS))
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--depth", type=int, default=0)
    ap.add_argument("--ppm", default="", help="also write the frame as a P3 file (main.cpp:245-255)")
    ap.add_argument("--tree-mode", type=int, default=0, help="1: binned-SAH mesh trees (free topology, no parity; SURVEY 8(f).2)")
    ap.add_argument("--mesh-depth", type=int, default=0)
    args = ap.parse_args()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X (no CPU fallback)")
    rt = importlib.import_module("ray-tracing_amd")
    name, td, md, W, H, spp, depth = CONFIGS[args.config]
    spp = args.spp or spp
    depth = args.depth or depth
    scene = rt.HostScene(name, td, args.mesh_depth or md, args.tree_mode)
    ctx = rt.Context(0)
    ctx.upload_scene(scene)
    ctx.set_camera(rt.default_camera(W / H))
    params = rt.make_params(W, H, spp=spp, max_depth=depth, light_samples=35, shadow_mode=0, jitter=1, seed=0)
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")

    tiles = [(x, y, min(32, W - x), min(32, H - y)) for y in range(0, H, 32) for x in range(0, W, 32)]

    def step():
        ctx.begin_frame(params)
        ctx.render_tiles(tiles)
        ctx.resolve_device(frame.data_ptr())
        return ctx.stats()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tot = dict(rays=0, primary=0, secondary=0, shadow=0, shadow_grouped=0, ms_shadow=0.0, ms_closest=0.0)
    for _ in range(args.steps):
        st = step()
        tot["rays"] += st.rays
        tot["primary"] += st.rays_primary
        tot["secondary"] += st.rays_secondary
        tot["shadow"] += st.rays_shadow
        tot["shadow_grouped"] += st.rays_shadow_grouped
        tot["ms_shadow"] += st.ms_shadow
        tot["ms_closest"] += st.ms_closest
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    k = args.steps
    print(json.dumps({
        "config": args.config, "scene": name, "tree_mode": args.tree_mode, "width": W, "height": H, "spp": spp, "max_depth": depth,
        "ms_per_frame": round(el / k * 1e3, 2), "mrays_per_s": round(tot["rays"] / el / 1e6, 2),
        "rays_per_frame": {c: tot[c] // k for c in ("primary", "secondary", "shadow", "shadow_grouped")},
        "kernel_ms_per_frame": {"k_shadow": round(tot["ms_shadow"] / k, 2), "k_closest": round(tot["ms_closest"] / k, 2)},
        "frame_checksum": float(frame.double().sum().item()),
    }), flush=True)
    if args.ppm:
        rt.write_ppm(args.ppm, frame.cpu().numpy())
    ctx.close()


if __name__ == "__main__":
    main()
