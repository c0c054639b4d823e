#!/usr/bin/env python3
"""tools/emu/dump_seed.py SEED DIR -- builds the scene of tests/test_gpu_parity.py::test_random_scenes[SEED], uploads it with
RT_DUMP_SCENE=DIR (the device scene as files: rtgpu.hip) and writes the origins and targets of shadow rays for tools/emu/shadow_emu
(DIR/rays.bin: n, then n x (P.xyz, L.xyz, radius) floats): the oracle's primary hits of the test's camera, towards both lights.
Runs no kernel: the upload only copies."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402


def build(seed, rt):
    r = np.random.default_rng(seed)
    hs, os_ = rt.HostScene(), O.Scene()
    meshes = []
    for k in range(int(r.integers(3, 7))):
        nv = int(r.integers(3, 400))
        scale = float(10.0 ** r.uniform(-1.5, 0.8))
        centre = r.uniform(-3, 3, 3)
        v = (centre + scale * r.normal(size=(nv, 3))).astype(np.float32)
        nt = int(r.integers(1, 600))
        t = r.integers(0, nv, (nt, 3)).astype(np.uint32)
        if k == 0:
            v[:3] = v[0] + np.array([[0, 0, 0], [1e-6, 0, 0], [0.5, 0.5e-6, 0]], np.float32) * scale
            t[0] = [0, 1, 2]
            t[1 % nt] = [0, 0, 1]
        meshes.append((v, t))
    spheres = [(r.uniform(-4, 4, 3).astype(np.float32), float(r.uniform(0.2, 1.5))) for _ in range(int(r.integers(0, 6)))]
    wall = np.array([[-8, -4, -7], [8, -4, -7], [8, 6, -7], [-8, 6, -7]], np.float32)
    floor = np.array([[-8, -4, 6], [8, -4, 6], [8, -4, -7], [-8, -4, -7]], np.float32)
    mesh_depth, top_depth = int(r.integers(2, 20)), int(r.integers(1, 5))
    for s, M in ((hs, rt), (os_, O)):
        def mat(kind, k):
            m = M.Material()
            c = np.random.default_rng(1000 * seed + k).uniform(0.2, 1.0, 3)
            m.kd[:] = c.tolist()
            m.ks[:] = c.tolist()
            m.shininess = 10.0
            m.type = kind
            m.ior = 1.3
            m.transparency = 0.5 if kind == 1 else 0.0
            m.reflectivity = 0.7 if kind == 2 else 0.0
            return m
        s.add_light([0.5, 4.0, 2.0], 0.4)
        s.add_light(meshes[0][0][0].tolist(), 0.2)
        s.add_square(wall, mat(0, 0))
        s.add_square(floor, mat(3, 1))
        for k, (c, rad) in enumerate(spheres):
            s.add_sphere(c, rad, mat(k % 3, 10 + k))
        for k, (v, t) in enumerate(meshes):
            s.add_mesh(v, t, mat(0 if k else 2, 20 + k))
        s.build(top_depth, mesh_depth)
    lights = [([0.5, 4.0, 2.0], 0.4), (meshes[0][0][0].tolist(), 0.2)]
    return hs, os_, lights, meshes, r


def main():
    seed, out = int(sys.argv[1]), sys.argv[2]
    os.makedirs(out, exist_ok=True)
    os.environ["RT_DUMP_SCENE"] = out
    rt = importlib.import_module("ray-tracing_amd")
    hs, os_, lights, meshes, r = build(seed, rt)
    ctx = rt.Context(0)
    ctx.upload_scene(hs)
    ctx.close()
    oc = O.default_camera(1.0, zoom=0.0, z=-9.0)
    co, cd = O.camera_rays(oc, O.make_params(56, 56, spp=1), 0, 0, 56, 56)
    h = os_.intersect(co, cd)
    pos = h["pos"][h["exists"] != 0]
    # plus points all over the meshes (bounced rays end there too)
    extra = np.concatenate([v[r.integers(0, len(v), 2000)] + r.normal(size=(2000, 3)).astype(np.float32) * 0.02 for v, _ in meshes])
    pts = np.concatenate([pos, extra]).astype(np.float32)
    rows = []
    for lp, rad in lights:
        rows.append(np.concatenate([pts, np.tile(np.asarray(lp, np.float32), (len(pts), 1)), np.full((len(pts), 1), rad, np.float32)], axis=1))
    rays = np.concatenate(rows).astype(np.float32)
    with open(os.path.join(out, "rays.bin"), "wb") as f:
        f.write(np.asarray([len(rays)], np.int64).tobytes())
        f.write(rays.tobytes())
    print("dumped", len(rays), "ray targets,", len(pos), "primary hits")


if __name__ == "__main__":
    main()
