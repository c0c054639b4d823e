#!/usr/bin/env python3
"""bench.py -- Mrays/s of the MI355X render loop on BASELINE config 5, ms/frame on the mesh scenes.

Workload (BASELINE.json configs[4]): synthetic 10 M-triangle soup + floor + one
area light, 3840x2160, 16 spp, soft shadows (35 samples per light), depth 5, mesh tree depth 24,
counter RNG seed 0.  One "step" = one full frame.  With --gpus N the frame's 32x32 tiles are dealt
to the N ranks (one process per GPU, scene replicated; ray-tracing_amd/dist.py) and the frame is
assembled with one RCCL all-gather of packed tile slabs -- strong scaling.

Prints ONE JSON line on rank 0 (the repo's bench contract):
  value        rays of all ranks / max-over-ranks wall time of the timed steps
  roofline     for the dominant kernel, k_shadow. It is bound on-chip, by two resources at once (experiments/README.md,
               "What binds k_shadow"): the vector L1's one lane-access per clock per CU (`l1`: the kernel's four
               16-byte loads per lane and node step sit on that ceiling) and, 15 % behind it, VALU issue. `bound` =
               "valu": `achieved` = executed VALU lane-operations per second (SQ_THREAD_CYCLES_VALU per ray from the PMC
               pass over this very frame, profiles/c5_k_shadow_pmc.json, x the rays of a launch / the launch duration
               measured live with HIP events inside the library), `peak` = the f32 vector peak of MI355X_MICROARCH.md
               (256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz = 78.6 T lane-ops/s); `slow_class_rate` is what the instruction
               classes the kernel mostly consists of can reach (half of it). Beside it:
               `hbm` (counter traffic vs 8 TB/s), `executed_bytes` (node and triangle records the walk requested
               vs the L2 peak) and `reference_algorithm_equiv_GBps` (SURVEY 8(d)'s bytes per ray of the
               reference's exhaustive traversal -- a speed-up figure, it exceeds the HBM peak because the GPU
               walk stops at the first occluder, prunes and reads quantised nodes; results are bit-identical).
  cpu_baseline the oracle (CPU restatement of the reference path, ThreadPool-equivalent) timed on this host on
               a centred 256x256 crop with 0.6 x cores (main.cpp:61) and with all cores of this process.
  config       also carries BASELINE's second half: ms/frame of the mesh scenes (C4 gorilla and cow), and C1-C3.
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
L2_PEAK_GBS = 34500.0          # ... aggregate L2 bandwidth
# VALU lane-operation peak: MI355X_MICROARCH.md prices a wave64 v_fma_f32 at 2 cycles per SIMD (f32 vector peak 157.3 TFLOP/s
# = 78.6 T lane-FMA/s = 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz). tools/micro/valu_rate.hip (profiles/r02_micro_valu_rate.txt)
# measures 66 T for v_add_u32, 53 T for v_fma_f32 and 35-38 T for the min/max/perm/cvt/cmp/select class (4 cycles per
# instruction) that most of k_shadow consists of: VALU_SLOW_CLASS is that second rate, reported beside the peak.
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12
VALU_PEAK_TFLOPS = 2.0 * VALU_PEAK_TLANEOPS       # 157.3 TFLOP/s f32 vector (an fma is two flops), MI355X_MICROARCH.md
VALU_SLOW_CLASS_TLANEOPS = 256 * 4 * 16 * 2.4e9 / 1e12
# Vector L1: one lane-access (one lane x one load instruction) per clock per CU for divergent loads, measured
# (tools/micro/l1_rate.hip, profiles/r02_micro_l1_rate.txt: 0.99 whatever the table size and the width of the load).
L1_LANE_ACCESS_PEAK_G = 256 * 1.0 * 2.4e9 / 1e9
PER_RAY_FILE = os.path.join(ROOT, "profiles", "c5_per_ray.json")
PMC_FILE = os.path.join(ROOT, "profiles", "c5_k_shadow_pmc.json")
PMC_C1_FILE = os.path.join(ROOT, "profiles", "c1_k_closest_pmc.json")
CHECKSUM_FILE = os.path.join(ROOT, "profiles", "c5_checksum.json")   # frame checksums of the plain reference-order traversal

SIDE_CONFIGS = {   # BASELINE configs 1-4: name -> (scene, width, height, spp, max_depth)
    "C1_default_scene_480_30spp": ("final", 480, 480, 30, 50),
    "C2_default_scene_480_16spp": ("final", 480, 480, 16, 50),
    "C3_refraction_1080p": ("refraction", 1920, 1080, 30, 5),
    "C4_mesh_gorilla_1080p": ("mesh_gorilla", 1920, 1080, 30, 50),
    "C4_mesh_cow_1080p": ("mesh_cow", 1920, 1080, 30, 50),
}


def algorithmic_bytes_per_ray(n_node, n_tri):
    """N_node*64 + N_tri*48 + 32 (ray in) + 16 (hit out); shadow rays fetch no normals."""
    return n_node * 64.0 + n_tri * 48.0 + 32.0 + 16.0


def kernel_source_hash():
    """sha256 over the kernel sources the library is built from; tools/pmc_c5_json.py stamps the PMC file with the same."""
    import hashlib
    h = hashlib.sha256()
    try:
        for f in ("ray-tracing_amd/csrc/rt_kernels.h", "ray-tracing_amd/csrc/rt_device.h", "ray-tracing_amd/csrc/rtgpu.hip"):
            h.update(open(os.path.join(ROOT, f), "rb").read())
    except OSError:
        return None
    return h.hexdigest()


def load_json(path):
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def closest_roofline(rays_per_frame, ms_per_frame):
    """The bounce path's fraction (C1: primaries, reflections, refractions down to depth 50): VALU lane-operations executed by
    k_closest per ray, from a committed PMC run of this build (tools/pmc_c1.sh -> profiles/c1_k_closest_pmc.json), times the rays of
    the frame just timed, over k_closest's time in it (HIP events on the library's stream), against the f32 vector peak."""
    pmc = load_json(PMC_C1_FILE)
    if not pmc or ms_per_frame <= 0:
        return None
    fresh = pmc.get("kernel_source_sha256") is not None and pmc.get("kernel_source_sha256") == kernel_source_hash()
    r = {"kernel": "k_closest", "bound": "valu", "unit": "Tlane-op/s", "peak": round(VALU_PEAK_TLANEOPS, 3), "pmc_matches_build": bool(fresh),
         "rays_per_frame": int(rays_per_frame), "kernel_ms_per_frame": round(ms_per_frame, 3)}
    if not fresh:
        r.update({"achieved": None, "frac": None, "note": "profiles/c1_k_closest_pmc.json was taken on other kernel sources: rerun tools/pmc_c1.sh + tools/pmc_c1_json.py"})
        return r
    achieved = pmc["valu_lane_ops_per_ray"] * rays_per_frame / (ms_per_frame * 1e-3) / 1e12
    r.update({"achieved": round(achieved, 3), "frac": round(achieved / VALU_PEAK_TLANEOPS, 4),
              "valu_lane_ops_per_ray": round(pmc["valu_lane_ops_per_ray"], 1), "lanes_active_per_valu_inst": round(pmc["lanes_active_per_valu_inst"], 4),
              "hbm": {"bytes_per_ray": round(pmc["hbm_bytes_per_ray"], 1), "achieved": round(pmc["hbm_bytes_per_ray"] * rays_per_frame / (ms_per_frame * 1e-3) / 1e9, 1),
                      "peak": HBM_PEAK_GBS, "unit": "GB/s"},
              "launches_per_frame": pmc["launches_per_frame"],
              "note": "the frame's k_closest time is mostly the latency of its launches on small batches (the depth-50 tail: 45 launches of 1 to 131 k states), not issue or bandwidth",
              "pmc_source": pmc.get("source")})
    return r


def time_side_config(rt, torch, device, name):
    """ms per frame of one of the 1080p configs on this GPU (1 warm-up frame, 3 timed)."""
    scene_name, W, H, spp, depth = SIDE_CONFIGS[name]
    scene = rt.HostScene(scene_name, 3, 12)
    ctx = rt.Context(device)
    try:
        ctx.upload_scene(scene)
        ctx.set_camera(rt.default_camera(W / H))
        params = rt.make_params(W, H, spp=spp, max_depth=depth, light_samples=35, shadow_mode=0, jitter=1, seed=0)
        frame = torch.zeros((H, W, 3), dtype=torch.float32, device=f"cuda:{device}")
        tiles = np.asarray(rt.interleaved_tiles(W, H), np.int32)

        def step():
            ctx.begin_frame(params)
            ctx.render_tiles(tiles)                    # the per-pixel tasks of main.cpp:210-225, queued with one call
            ctx.resolve_device(frame.data_ptr())
            return ctx.stats()

        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rays = 0
        k = 3
        ms_closest = ms_shadow = 0.0
        for _ in range(k):
            st = step()
            rays += st.rays
            ms_closest += st.ms_closest
            ms_shadow += st.ms_shadow
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        st = ctx.stats()
        out = {"ms_per_frame": round(el / k * 1e3, 2), "mrays_per_s": round(rays / el / 1e6, 1), "spp": spp, "max_depth": depth,
               "rays_traced_per_frame": st.rays, "shadow_rays_culled_per_frame": st.rays_shadow_culled,
               "shadow_queries_settled_by_group_test_per_frame": st.rays_shadow_grouped,       # answered without a ray of their own; not in rays_traced, not in mrays_per_s
               "kernel_ms_per_frame": {"k_shadow": round(ms_shadow / k, 2), "k_closest": round(ms_closest / k, 2)},
               "frame_checksum": float(np.nan_to_num(frame.double().cpu().numpy(), nan=0.0, posinf=0.0, neginf=0.0).sum())}
        if name.startswith("C1"):
            out["roofline_k_closest"] = closest_roofline(st.rays_primary + st.rays_secondary, ms_closest / k)
        return out
    finally:
        ctx.close()


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
    ap.add_argument("--cpu-crop", type=int, default=256, help="side of the centred crop the CPU baseline renders (SURVEY 8(d): 256)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the C3/C4 frames")
    ap.add_argument("--shard", default="tiles", choices=["tiles", "samples"], help="how N ranks split the frame: tiles (default: disjoint 32x32 tiles, one all-gather of packed slabs) or samples (every rank renders its samples of every pixel, one all-reduce of the fixed-point accumulators: for frames in which a few pixels dominate, SURVEY 8(e))")
    ap.add_argument("--tree-mode", type=int, default=0, help="1: binned-SAH mesh tree, 2: linear BVH built on the GPU (free topology: NOT the reference's answers, SURVEY 8(f).2); never the default")
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
    # rehearsal knobs for a one-GPU box (never set by the driver): all ranks on one card, collectives over gloo
    if os.environ.get("RT_BENCH_ONE_DEVICE"):
        local_rank = int(os.environ["RT_BENCH_ONE_DEVICE"])
    torch.cuda.set_device(local_rank)
    rt = importlib.import_module("ray-tracing_amd")
    rtd = importlib.import_module("ray-tracing_amd.dist")
    rtd.init(os.environ.get("RT_BENCH_BACKEND", "nccl"), local_rank)          # RCCL; no-op for one rank

    W, H = args.width, args.height
    name = f"soup:{args.ntris}"
    key = f"{name}/d{args.mesh_depth}/{W}x{H}x{args.spp}"
    t0 = time.time()
    scene = rt.HostScene(name, 3, args.mesh_depth, args.tree_mode)
    t_build = time.time() - t0
    ctx = rt.Context(local_rank)
    t0 = time.time()
    ctx.upload_scene(scene)
    t_upload = time.time() - t0
    cam = rt.default_camera(W / H)
    ctx.set_camera(cam)
    params = rt.make_params(W, H, spp=args.spp, max_depth=args.depth, light_samples=35, shadow_mode=0, jitter=1, seed=0)
    dev = f"cuda:{local_rank}"
    by_samples = args.shard == "samples" and world > 1
    tiles = rtd.shard_tiles(W, H, 0, 1, 32) if by_samples else rtd.shard_tiles(W, H, rank, world, 32)
    gather = rtd.FrameGather(W, H, world, dev, 32)
    reduce_samples = rtd.SampleReduce(W, H, world, dev) if by_samples else None
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)

    tiles = np.asarray(tiles, np.int32).reshape(-1, 4)

    def step():
        if by_samples:
            ctx.begin_frame_samples(params, *rtd.sample_range(rank, world, args.spp))
            ctx.render_tiles(tiles)                # every tile, this rank's samples
            reduce_samples(ctx)                    # RCCL all-reduce(sum) of the int64 accumulators (+ flag words)
            ctx.resolve_device(frame.data_ptr())   # every rank resolves the same sums: the whole frame
            return ctx.stats()
        ctx.begin_frame(params)
        ctx.render_tiles(tiles)                    # this rank's tiles, queued with one call
        ctx.resolve_device(frame.data_ptr())       # runs the queued tiles, writes this rank's pixels
        gather(frame, rank)                        # RCCL all-gather of the packed tile slabs + scatter
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
    elapsed = rtd.max_over_ranks(elapsed, dev)
    my_shad = shad
    rays, prim, sec, shad = rtd.sum_counters([rays, prim, sec, shad], dev)
    checksum = float(frame.double().sum().item())

    if rank == 0:
        # ---- BASELINE's second half: ms/frame on the mesh scenes (and C3), same kernels, this GPU ----
        side = None
        if world == 1 and not args.no_side:
            side = {}
            for cfg in SIDE_CONFIGS:
                try:
                    side[cfg] = time_side_config(rt, torch, local_rank, cfg)
                except Exception as e:      # a side frame must not cost the headline line
                    side[cfg] = {"error": str(e)[:200]}
        # ---- CPU baseline + per-ray algorithmic work, from the oracle (test infrastructure) ----
        cpu = None
        per_ray = (load_json(PER_RAY_FILE) or {}).get(key)
        if world == 1 and not args.no_cpu and args.tree_mode == 0:
            import oracle_lib as O
            avail = len(os.sched_getaffinity(0))
            if os.environ.get("OMP_NUM_THREADS", "").isdigit():
                avail = max(1, min(avail, int(os.environ["OMP_NUM_THREADS"])))     # the GPU box grants 16 cores per GPU
            avail = min(avail, 64)
            t_ref = max(1, int(avail * 0.6))                                        # main.cpp:61,204
            osc = O.Scene(name, 3, args.mesh_depth)
            oc = O.default_camera(W / H)
            op = O.make_params(W, H, spp=args.spp, max_depth=args.depth, light_samples=35, shadow_mode=0, jitter=1, seed=0)
            c = min(args.cpu_crop, W, H)
            x0, y0 = (W - c) // 2, (H - c) // 2
            runs = []
            q = ost = orays = None
            for threads in sorted({t_ref, avail}):
                orgb, oprim, ost, secs = osc.render(oc, op, x0, y0, c, c, threads=threads)
                orays = ost.rays_primary + ost.rays_secondary + ost.rays_shadow
                runs.append({"threads": threads, "value": round(orays / secs / 1e6, 4), "seconds": round(secs, 1)})
                if q is None:
                    got = frame[y0:y0 + c, x0:x0 + c].cpu().numpy()
                    q = int(np.abs(O.quantize(got) - O.quantize(orgb)).max())
            cpu = {"value": runs[-1]["value"], "unit": "Mrays/s", "cores": runs[-1]["threads"], "kind": "port",
                   "runs": runs, "host_cpus_visible": os.cpu_count(), "cores_of_this_process": avail,
                   "sample": f"centred {c}x{c} crop of the {W}x{H} frame at {args.spp} spp at the full-resolution camera ({orays} rays per run); "
                             f"per-pixel tasks on {runs[0]['threads']} threads (0.6 x cores, the reference's default) and on {runs[-1]['threads']} (all cores of this process); "
                             f"GPU frame vs oracle on the crop: max 8-bit delta {q}"}
            per_ray = {"shadow_nodes": ost.sh_node_visits / max(ost.rays_shadow, 1),
                       "shadow_tris": ost.sh_tri_tests / max(ost.rays_shadow, 1),
                       "closest_nodes": (ost.node_visits - ost.sh_node_visits) / max(ost.rays_primary + ost.rays_secondary, 1),
                       "closest_tris": (ost.tri_tests - ost.sh_tri_tests) / max(ost.rays_primary + ost.rays_secondary, 1)}
        want = (load_json(CHECKSUM_FILE) or {}).get(f"{key}/depth{args.depth}")
        checksum_ok = None if want is None else bool(want == checksum)
        roof = None
        pmc = (load_json(PMC_FILE) or {}).get(key)
        if n_shadow:
            rays_per_launch = my_shad / n_shadow
            avg_s = ms_shadow / n_shadow * 1e-3
            roof = {"kernel": "k_shadow", "avg_launch_ms": round(avg_s * 1e3, 3), "launches": n_shadow, "rays_per_launch": round(rays_per_launch, 1),
                    "closest_kernel_ms": round(ms_closest / max(n_closest, 1), 3)}
            if pmc:
                # the per-ray counters come from a committed PMC run (tools/pmc_c5.sh): they describe THIS build only if the kernel
                # sources have not changed since -- otherwise no fraction is quoted
                fresh = pmc.get("kernel_source_sha256") is not None and pmc.get("kernel_source_sha256") == kernel_source_hash()
                roof["pmc_matches_build"] = bool(fresh)
            if pmc and not roof["pmc_matches_build"]:
                roof.update({"bound": "valu", "achieved": None, "peak": round(VALU_PEAK_TLANEOPS, 3), "unit": "Tlane-op/s", "frac": None, "traffic": None,
                             "note": "profiles/c5_k_shadow_pmc.json was taken on other kernel sources: rerun tools/pmc_c5.sh + tools/pmc_c5_json.py"})
            elif pmc:
                ops_per_ray = pmc["SQ_THREAD_CYCLES_VALU_per_launch"] / pmc["rays_per_launch"]
                achieved = ops_per_ray * rays_per_launch / avg_s / 1e12
                # SURVEY 8(d)'s flop formula on the steps the kernel actually executed (RT_COUNT of the PMC run): 4 boxes per 4-wide node step
                flops_per_ray = pmc["node_steps_per_ray"] * 4.0 * 30.0 + pmc["tri_steps_per_ray"] * 160.0
                hbm_bytes = (2.0 * pmc["FETCH_SIZE_KB_per_launch"] + pmc["WRITE_SIZE_KB_per_launch"]) * 1024.0 / pmc["rays_per_launch"] * rays_per_launch
                ex_bpr = pmc["node_steps_per_ray"] * 64.0 + pmc["tri_steps_per_ray"] * 128.0 + 32.0
                roof.update({
                    "bound": "valu", "achieved": round(achieved, 3), "peak": round(VALU_PEAK_TLANEOPS, 3), "unit": "Tlane-op/s",
                    "frac": round(achieved / VALU_PEAK_TLANEOPS, 4), "traffic": round(hbm_bytes),
                    "flop_frac": round(flops_per_ray * rays_per_launch / avg_s / 1e12 / VALU_PEAK_TFLOPS, 4),
                    "flops_per_ray_8d_formula_executed_steps": round(flops_per_ray, 1),
                    "valu_lane_ops_per_ray": round(ops_per_ray, 1),
                    "lanes_active_per_valu_inst": round(pmc["SQ_THREAD_CYCLES_VALU_per_launch"] / (64.0 * pmc["SQ_INSTS_VALU_per_launch"]), 4),
                    "valu_insts_per_ray": round(pmc["SQ_INSTS_VALU_per_launch"] / pmc["rays_per_launch"], 1),
                    "frac_of_slow_class_rate": round(achieved / VALU_SLOW_CLASS_TLANEOPS, 4),
                    "slow_class_rate": round(VALU_SLOW_CLASS_TLANEOPS, 3),
                    # the co-limit: k_shadow's four 16-byte loads per lane and node step against the vector L1's one lane-access per clock
                    "l1": {"lane_accesses_per_ray": round(pmc["TCP_TOTAL_CACHE_ACCESSES_sum_per_launch"] / pmc["rays_per_launch"], 1),
                           "achieved": round(pmc["TCP_TOTAL_CACHE_ACCESSES_sum_per_launch"] / pmc["rays_per_launch"] * rays_per_launch / avg_s / 1e9, 1),
                           "ceiling": round(L1_LANE_ACCESS_PEAK_G, 1), "unit": "G lane-accesses/s",
                           "ratio": round(pmc["TCP_TOTAL_CACHE_ACCESSES_sum_per_launch"] / pmc["rays_per_launch"] * rays_per_launch / avg_s / 1e9 / L1_LANE_ACCESS_PEAK_G, 4),
                           "note": "ceiling measured for divergent loads (tools/micro/l1_rate.hip); lanes of one instruction that want the same line are served together, so a kernel with partly coherent lanes can read a little above it"},
                    "hbm": {"achieved": round(hbm_bytes / avg_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(hbm_bytes / avg_s / 1e9 / HBM_PEAK_GBS, 4)},
                    "executed_bytes": {"bytes_per_ray": round(ex_bpr, 1), "achieved": round(ex_bpr * rays_per_launch / avg_s / 1e9, 1),
                                       "peak": L2_PEAK_GBS, "unit": "GB/s (L2)", "frac": round(ex_bpr * rays_per_launch / avg_s / 1e9 / L2_PEAK_GBS, 4)},
                    # what north_star asks the counters to show beside the fractions: pipe utilisation, occupancy, cache hit rates
                    "valu_busy": round(pmc["waves_per_simd"] * pmc["SQ_ACTIVE_INST_VALU_per_launch"] / pmc["SQ_WAVE_CYCLES_per_launch"], 4),
                    "waves_per_simd": pmc["waves_per_simd"],
                    "wave_cycles_waiting": round(pmc["SQ_WAIT_ANY_per_launch"] / pmc["SQ_WAVE_CYCLES_per_launch"], 4),
                    "l2_hit": round(pmc["TCC_HIT_sum_per_launch"] / (pmc["TCC_HIT_sum_per_launch"] + pmc["TCC_MISS_sum_per_launch"]), 4),
                    "l1_hit": round(1.0 - pmc["TCP_TCC_READ_REQ_sum_per_launch"] / pmc["TCP_TOTAL_CACHE_ACCESSES_sum_per_launch"], 4),
                    "lanes_active_node_phase": round(pmc["node_phase_lanes_active"], 4), "lanes_active_triangle_phase": round(pmc["tri_phase_lanes_active"], 4),
                    "pmc_source": pmc.get("source")})
            if per_ray:
                bpr = algorithmic_bytes_per_ray(per_ray["shadow_nodes"], per_ray["shadow_tris"])
                roof["reference_algorithm_bytes_per_ray"] = round(bpr, 1)
                roof["reference_algorithm_equiv_GBps"] = round(bpr * rays_per_launch / avg_s / 1e9, 1)
        cfg = {"tree_mode": args.tree_mode, "workload": f"C5: synthetic {args.ntris}-triangle soup, {W}x{H}, {args.spp} spp, soft shadows "
                           f"(35 samples/light), depth {args.depth}, mesh BVH depth {args.mesh_depth}, counter RNG",
               "tiles": "32x32, round-robin per tile row with a hashed row offset", "parallelism": f"{'sample' if by_samples else 'tile'}-sharded x{world}",
               "gather": (f"all-reduce of the fixed-point accumulators, {reduce_samples.bytes_reduced_per_rank()} B per rank" if by_samples else
                          f"all-gather of packed tile slabs, {gather.bytes_sent_per_rank()} B per rank") if world > 1 else "none (one rank)",
               "rays_per_frame": {"primary": prim // args.steps, "secondary": sec // args.steps, "shadow": shad // args.steps},
               "kernel_ms_per_step": {"k_shadow": round(ms_shadow / args.steps, 2), "k_closest": round(ms_closest / args.steps, 2)},
               "scene_build_s": round(t_build, 2), "upload_s": round(t_upload, 2), "frame_checksum": checksum,
               "frame_checksum_matches_reference_order_traversal": checksum_ok}
        if side:
            cfg["ms_per_frame_mesh_scene"] = {k: v.get("ms_per_frame") for k, v in side.items() if k.startswith("C4")}
            cfg["side_configs"] = side
        out = {
            "metric": "Mrays/sec (primary+secondary+shadow); ms/frame on mesh scene in config", "value": round(rays / elapsed / 1e6, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": cfg, "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
