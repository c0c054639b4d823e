#!/usr/bin/env python3
"""bench.py -- Mrays/s of the MI355X render loop on BASELINE config 5.

Workload (BASELINE.json configs[4]): synthetic 10 M-triangle soup + floor + one
area light, 3840x2160, 16 spp, soft shadows (35 samples per light), depth 5, mesh tree depth 24,
counter RNG seed 0.  One "step" = one full frame.  With --gpus N the frame's 32x32 tiles are dealt
round-robin to the N ranks (one process per GPU, scene replicated) and the frame is assembled
with one RCCL all-reduce (disjoint tiles, so the sum is a gather) -- strong scaling.

Prints ONE JSON line on rank 0 (see the repo's bench contract): value = rays of all ranks /
max-over-ranks wall time; `roofline` for the dominant kernel (k_shadow) from HIP-event launch
durations measured inside the library; `cpu_baseline` = the oracle (CPU restatement of the
reference path, ThreadPool-equivalent) timed on this host on a bounded centred crop.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PER_RAY_FILE = os.path.join(ROOT, "profiles", "c5_per_ray.json")
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "hbm_traffic.json")
CHECKSUM_FILE = os.path.join(ROOT, "profiles", "c5_checksum.json")   # frame checksums of the plain reference-order traversal


def algorithmic_bytes_per_ray(n_node, n_tri):
    """N_node*64 + N_tri*48 + 32 (ray in) + 16 (hit out); shadow rays fetch no normals."""
    return n_node * 64.0 + n_tri * 48.0 + 32.0 + 16.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ntris", type=int, default=10_000_000)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--mesh-depth", type=int, default=24)
    ap.add_argument("--cpu-crop", type=int, default=64, help="side of the centred crop the CPU baseline renders")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    rt = importlib.import_module("ray-tracing_amd")
    rtd = importlib.import_module("ray-tracing_amd.dist")
    rtd.init("nccl", local_rank)          # RCCL; no-op for one rank

    W, H = args.width, args.height
    name = f"soup:{args.ntris}"
    t0 = time.time()
    scene = rt.HostScene(name, 3, args.mesh_depth)
    t_build = time.time() - t0
    ctx = rt.Context(local_rank)
    t0 = time.time()
    ctx.upload_scene(scene)
    t_upload = time.time() - t0
    cam = rt.default_camera(W / H)
    ctx.set_camera(cam)
    params = rt.make_params(W, H, spp=args.spp, max_depth=args.depth, light_samples=35, shadow_mode=0, jitter=1, seed=0)
    tiles = rtd.shard_tiles(W, H, rank, world, 32)
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device=f"cuda:{local_rank}")

    def step():
        ctx.begin_frame(params)
        for t in tiles:
            ctx.render_tile(*t)
        ctx.resolve_device(frame.data_ptr())       # runs the queued tiles, writes this rank's pixels
        rtd.gather_frame(frame)                    # RCCL all-reduce: disjoint tiles -> gather
        return ctx.stats()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    rays = prim = sec = shad = 0
    ms_shadow = ms_closest = 0.0
    n_shadow = n_closest = 0
    for _ in range(args.steps):
        st = step()
        rays += st.rays
        prim += st.rays_primary
        sec += st.rays_secondary
        shad += st.rays_shadow
        ms_shadow += st.ms_shadow
        ms_closest += st.ms_closest
        n_shadow += st.launches_shadow
        n_closest += st.launches_closest
    fence()
    elapsed = time.perf_counter() - t0
    dev = f"cuda:{local_rank}"
    elapsed = rtd.max_over_ranks(elapsed, dev)
    rays, prim, sec, shad = rtd.sum_counters([rays, prim, sec, shad], dev)
    checksum = float(frame.double().sum().item())

    if rank == 0:
        # ---- CPU baseline + per-ray algorithmic work, from the oracle (test infrastructure) ----
        cpu = None
        per_ray = None
        if os.path.exists(PER_RAY_FILE):
            try:
                per_ray = json.load(open(PER_RAY_FILE)).get(f"{name}/d{args.mesh_depth}/{W}x{H}x{args.spp}")
            except Exception:
                per_ray = None
        if world == 1 and not args.no_cpu:
            import oracle_lib as O
            threads = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16")), 64))
            osc = O.Scene(name, 3, args.mesh_depth)
            oc = O.default_camera(W / H)
            op = O.make_params(W, H, spp=args.spp, max_depth=args.depth, light_samples=35, shadow_mode=0, jitter=1, seed=0)
            c = args.cpu_crop
            x0, y0 = (W - c) // 2, (H - c) // 2
            orgb, oprim, ost, secs = osc.render(oc, op, x0, y0, c, c, threads=threads)
            orays = ost.rays_primary + ost.rays_secondary + ost.rays_shadow
            got = frame[y0:y0 + c, x0:x0 + c].cpu().numpy()
            q = np.abs(O.quantize(got) - O.quantize(orgb)).max()
            cpu = {"value": round(orays / secs / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
                   "sample": f"centred {c}x{c} crop of the {W}x{H} frame at {args.spp} spp ({orays} rays, {secs:.1f} s); "
                             f"GPU frame vs oracle on the crop: max 8-bit delta {int(q)}"}
            per_ray = {"shadow_nodes": ost.sh_node_visits / max(ost.rays_shadow, 1),
                       "shadow_tris": ost.sh_tri_tests / max(ost.rays_shadow, 1),
                       "closest_nodes": (ost.node_visits - ost.sh_node_visits) / max(ost.rays_primary + ost.rays_secondary, 1),
                       "closest_tris": (ost.tri_tests - ost.sh_tri_tests) / max(ost.rays_primary + ost.rays_secondary, 1)}
        checksum_ok = None
        try:
            want = json.load(open(CHECKSUM_FILE)).get(f"{name}/d{args.mesh_depth}/{W}x{H}x{args.spp}/depth{args.depth}")
            if want is not None:
                checksum_ok = bool(want == checksum)
        except Exception:
            pass
        roof = None
        if per_ray and n_shadow:
            bpr = algorithmic_bytes_per_ray(per_ray["shadow_nodes"], per_ray["shadow_tris"])
            rays_per_launch = shad / world / n_shadow if world > 1 else shad / n_shadow
            avg_ms = ms_shadow / n_shadow
            achieved = bpr * rays_per_launch / (avg_ms * 1e-3) / 1e9
            traffic = None
            if os.path.exists(TRAFFIC_FILE):
                try:
                    traffic = json.load(open(TRAFFIC_FILE)).get("k_shadow_bytes_per_launch")
                except Exception:
                    traffic = None
            traffic_rate = round(traffic / (avg_ms * 1e-3) / 1e9, 1) if traffic else None     # measured HBM-side GB/s
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_rate": traffic_rate,
                    "traffic_frac": round(traffic_rate / HBM_PEAK_GBS, 4) if traffic_rate else None, "kernel": "k_shadow",
                    "avg_launch_ms": round(avg_ms, 3), "launches": n_shadow,
                    "bytes_per_ray": round(bpr, 1), "rays_per_launch": round(rays_per_launch, 1),
                    "closest_kernel_ms": round(ms_closest / max(n_closest, 1), 3)}
        out = {
            "metric": "Mrays/sec (primary+secondary+shadow)", "value": round(rays / elapsed / 1e6, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C5: synthetic {args.ntris}-triangle soup, {W}x{H}, {args.spp} spp, soft shadows "
                                   f"(35 samples/light), depth {args.depth}, mesh BVH depth {args.mesh_depth}, counter RNG",
                       "tiles": "32x32 round-robin over ranks", "parallelism": f"tile-sharded x{world}",
                       "rays_primary": prim, "rays_secondary": sec, "rays_shadow": shad,
                       "kernel_ms_per_step": {"k_shadow": round(ms_shadow / args.steps, 2), "k_closest": round(ms_closest / args.steps, 2)},
                       "scene_build_s": round(t_build, 2), "upload_s": round(t_upload, 2), "frame_checksum": checksum, "frame_checksum_matches_reference_order_traversal": checksum_ok},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
