#!/usr/bin/env python3
"""tools/emu/dump_named.py NAME MESH_DEPTH DIR -- as dump_seed.py for one of the named scenes (rth::Scene::setup): the device scene as files plus
ray targets for tools/emu/shadow_emu (random points of the scene's neighbourhood towards its lights). Runs no kernel."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402

name, md, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
os.makedirs(out, exist_ok=True)
os.environ["RT_DUMP_SCENE"] = out
rt = importlib.import_module("ray-tracing_amd")
hs = rt.HostScene(name, 3, md)
ctx = rt.Context(0)
ctx.upload_scene(hs)
ctx.close()
os_ = O.Scene(name, 3, md)
oc = O.default_camera(1.0)
co, cd = O.camera_rays(oc, O.make_params(64, 64, spp=1), 0, 0, 64, 64)
h = os_.intersect(co, cd)
pos = h["pos"][h["exists"] != 0]
r = np.random.default_rng(1)
pts = np.concatenate([pos, r.uniform(-3, 3, (3000, 3)).astype(np.float32)]).astype(np.float32)
lights = [([0.0, 2.5, 0.0], 0.5), ([1.0, 8.0, -1.0], 2.0)]
rows = [np.concatenate([pts, np.tile(np.asarray(lp, np.float32), (len(pts), 1)), np.full((len(pts), 1), rad, np.float32)], axis=1) for lp, rad in lights]
rays = np.concatenate(rows).astype(np.float32)
with open(os.path.join(out, "rays.bin"), "wb") as f:
    f.write(np.asarray([len(rays)], np.int64).tobytes())
    f.write(rays.tobytes())
print("dumped", name, len(rays), "ray targets")
