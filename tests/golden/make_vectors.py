#!/usr/bin/env python3
"""Freezes oracle outputs into tests/golden/vectors/*.npz (run from the repo root: python tests/golden/make_vectors.py).

One file per BASELINE config, small centred crops at the config's own camera and image size:
  * closest hits of the pixel-centre primary rays: exists / object ordinal / triangle index and the BITS of t, position,
    normal, u, v (Scene::computeIntersection);
  * radiance BITS of Scene::rayTrace on those rays with hard shadows (RNG-free) and with soft shadows (counter RNG);
  * ray counts, and the node / leaf / empty-leaf / largest-leaf statistics of the top-level tree and of mesh 0.
tests/test_golden_vectors.py holds the oracle (CPU) and the HIP path (GPU) against these files, so that the two cannot
drift together. The vectors come from this repository's oracle, not from the reference binary (unbuildable here:
DESIGN.md section 6): they freeze behaviour, they do not lift "parity unpinned".
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

OUT = os.path.join(HERE, "vectors")

# name: scene, mesh depth, image w, h, crop side, camera zoom, max_depth, meshes to record
CONFIGS = {
    "C1_final_default_pose": dict(scene="final", mesh_depth=12, w=480, h=480, crop=40, zoom=3.0, max_depth=50),
    "C1_final_pulled_back": dict(scene="final", mesh_depth=12, w=480, h=480, crop=40, zoom=25.0, max_depth=6),
    # glass meshes in direct view at the reference's own depth (Scene.h:283-286): what contribution culling and the LIFO stack are
    # held against. Minutes of oracle time (the depth-50 recursion inside the glass flamingos), so only the GPU test loads it.
    # ("window": the crop x crop grid of rays and the rendered crop sit on the glass gorilla's place in the frame, pixels 90-180 x 165-285)
    "C1_final_pulled_back_d50": dict(scene="final", mesh_depth=12, w=480, h=480, crop=12, zoom=25.0, max_depth=50, window=[90, 165, 90, 120]),
    "C3_refraction_1080p": dict(scene="refraction", mesh_depth=12, w=1920, h=1080, crop=64, zoom=3.0, max_depth=5),
    "C4_mesh_gorilla_1080p": dict(scene="mesh_gorilla", mesh_depth=12, w=1920, h=1080, crop=64, zoom=3.0, max_depth=50),
    "C4_mesh_cow_1080p": dict(scene="mesh_cow", mesh_depth=12, w=1920, h=1080, crop=64, zoom=3.0, max_depth=50),
    "F1_reflexion_chamber": dict(scene="reflexion_chamber", mesh_depth=12, w=480, h=480, crop=48, zoom=3.0, max_depth=50),
    "F1_cornell_box": dict(scene="cornell_box", mesh_depth=12, w=480, h=480, crop=48, zoom=3.0, max_depth=50),
    "C5_soup_200k_4k": dict(scene="soup:200000", mesh_depth=18, w=3840, h=2160, crop=48, zoom=3.0, max_depth=5),
    # the benched scene itself; its oracle takes a while to build, so only the GPU test loads this one
    "C5_soup_10M_4k": dict(scene="soup:10000000", mesh_depth=24, w=3840, h=2160, crop=32, zoom=3.0, max_depth=5),
}


def vector_inputs(cfg):
    """The rays and parameters of a config (shared with the test): pixel-centre primary rays on a crop x crop grid of
    pixels spread over the whole frame (so that every object class of the scene is seen), and the centred crop rectangle
    the render entry point is run on."""
    w, h, c = cfg["w"], cfg["h"], cfg["crop"]
    cam = O.default_camera(w / h, zoom=cfg["zoom"])
    wx, wy, ww, wh = cfg.get("window", [0, 0, w, h])
    x0, y0 = wx + (ww - c) // 2, wy + (wh - c) // 2
    p1 = O.make_params(w, h, spp=1, max_depth=cfg["max_depth"], light_samples=35, shadow_mode=1, jitter=0, seed=11)
    xs = wx + ((np.arange(c) * 2 + 1) * ww) // (2 * c)
    ys = wy + ((np.arange(c) * 2 + 1) * wh) // (2 * c)
    org, dirs = [], []
    for y in ys:
        o, d = O.camera_rays(cam, p1, 0, int(y), w, 1)
        org.append(o[xs])
        dirs.append(d[xs])
    org, dirs = np.concatenate(org), np.concatenate(dirs)
    pixel = (xs[None, :] + ys[:, None] * w).reshape(-1).astype(np.uint32)
    return cam, x0, y0, org, dirs, pixel


def trace_parallel(sc, org, dirs, p, pixel, workers=8):
    """Scene::rayTrace per ray, the rays dealt to a few threads (every ray is on its own: same bits as one call)."""
    from concurrent.futures import ThreadPoolExecutor
    n = len(org)
    if n < 2 * workers:
        return sc.trace(org, dirs, p, pixel)
    out = np.zeros((n, 3), np.float32)
    parts = [np.arange(k, n, workers) for k in range(workers)]
    with ThreadPoolExecutor(workers) as ex:
        for idx, rgb in zip(parts, ex.map(lambda idx: sc.trace(org[idx], dirs[idx], p, pixel[idx]), parts)):
            out[idx] = rgb
    return out


def tree_stats(sc, which):
    leaf, cnt, _, items = sc.tree(which)
    return [len(leaf), int(leaf.sum()), int(((leaf == 1) & (cnt == 0)).sum()), int(cnt.max()) if len(cnt) else 0, len(items)]


def make(name, cfg):
    sc = O.Scene(cfg["scene"], 3, cfg["mesh_depth"])
    cam, x0, y0, org, dirs, pixel = vector_inputs(cfg)
    hits = sc.intersect(org, dirs)
    out = {"config": np.frombuffer(json.dumps(cfg).encode(), np.uint8)}
    for f in ("exists", "obj", "tri"):
        out["hit_" + f] = hits[f].astype(np.int32)
    for f in ("t", "pos", "nrm", "u", "v"):
        out["hit_" + f + "_bits"] = np.ascontiguousarray(hits[f]).view(np.uint32)
    for mode, tag in ((1, "hard"), (0, "soft")):
        p = O.make_params(cfg["w"], cfg["h"], spp=1, max_depth=cfg["max_depth"], light_samples=35, shadow_mode=mode, jitter=0, seed=11)
        out["rgb_" + tag + "_bits"] = trace_parallel(sc, org, dirs, p, pixel).view(np.uint32)
        print(name, tag, "traced", flush=True)
    # ray counts of the same crop through the render entry point (pixel centres, hard shadows)
    p = O.make_params(cfg["w"], cfg["h"], spp=1, max_depth=cfg["max_depth"], light_samples=35, shadow_mode=1, jitter=0, seed=11)
    rgb, prim, st, _ = sc.render(cam, p, x0, y0, cfg["crop"], cfg["crop"], threads=8)
    out["render_rgb_bits"] = rgb.view(np.uint32)
    out["render_prim"] = prim.astype(np.int32)
    out["rays"] = np.array([st.rays_primary, st.rays_secondary, st.rays_shadow], np.int64)
    nm = sc.counts()[2]
    out["tree_top"] = np.array(tree_stats(sc, -1), np.int64)
    out["tree_mesh0"] = np.array(tree_stats(sc, 0) if nm else [0] * 5, np.int64)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, {k: (v.shape, v.dtype.name) for k, v in out.items() if k != "config"}, flush=True)


if __name__ == "__main__":
    for name, cfg in CONFIGS.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        make(name, cfg)
