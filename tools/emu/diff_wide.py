#!/usr/bin/env python3
"""tools/emu/diff_wide.py SEED [mode] -- a frame of build_wide(SEED) on the GPU against the oracle's, pixel by pixel, and against the GPU's own
frames under the alternative paths (exact shadow nodes, no culling, no group test, no front pass): which component deviates."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools", "emu"))
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402
from dump_seed import build_wide  # noqa: E402

seed = int(sys.argv[1])
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rt = importlib.import_module("ray-tracing_amd")
hs, os_, lights, meshes, r = build_wide(seed, rt)
print("lights", lights)
gc, oc = rt.default_camera(1.0, zoom=0.0, z=-10.0), O.default_camera(1.0, zoom=0.0, z=-10.0)
gp = rt.make_params(32, 32, spp=1, max_depth=5, jitter=1, shadow_mode=mode, seed=seed)
op = O.make_params(32, 32, spp=1, max_depth=5, jitter=1, shadow_mode=mode, seed=seed)
orgb, oprim, ost, _ = os_.render(oc, op, threads=16)


def frame(env):
    for k, v in env.items():
        os.environ[k] = v
    c = rt.Context(0)
    try:
        c.upload_scene(hs)
        rgb, prim, st = c.render(gc, gp)
    finally:
        c.close()
        for k in env:
            del os.environ[k]
    return rgb, st


base, st = frame({})
q = lambda a: O.quantize(np.nan_to_num(a, nan=0.0))
d = np.abs(q(base) - q(orgb)).max(axis=-1)
bad = np.argwhere(d > 1)
print("pixels off by more than a level:", len(bad), "rays", st.rays_primary, st.rays_secondary, st.rays_shadow, "oracle", ost.rays_primary, ost.rays_secondary, ost.rays_shadow)
for y, x in bad[:12]:
    print((int(y), int(x)), "gpu", base[y, x], "oracle", orgb[y, x], "prim", oprim[y, x])
for name, env in (("exact shadow nodes", {"RT_SHADOW_EXACT_NODES": "1"}), ("no culling", {"RT_NO_CULL": "1"}), ("no group test", {"RT_GROUP_CULL": "0"}),
                  ("no front pass", {"RT_FRONT": "0"}), ("front pass forced", {"RT_FRONT": "1"}), ("no leaf sub-trees", {"RT_NO_LEAF_SPLIT": "1"}),
                  ("no object sub-trees", {"RT_NO_TOP_SPLIT": "1"}), ("no flat pass", {"RT_NO_FLAT_TOP": "1"}), ("binary closest", {"RT_WIDE_CLOSEST": "0"})):
    f, _ = frame(env)
    same = np.array_equal(np.nan_to_num(f, nan=-1.0).view(np.uint32), np.nan_to_num(base, nan=-1.0).view(np.uint32))
    dd = np.abs(q(f) - q(orgb)).max(axis=-1)
    print(f"{name:22s}: same bits as the default frame: {same}; pixels off against the oracle: {int((dd > 1).sum())}")
