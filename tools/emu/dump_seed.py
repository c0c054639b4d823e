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


def build_wide(seed, rt, mirrors=False, plain=False):
    """A second family of random scenes for the soak runs: triangle soups of a few thousand small triangles with a few huge ones mixed in,
    tree depths 2..22, up to 40 spheres (many-object top levels: batched top-level rounds, sub-trees over objects, the flat pass), one to four
    lights of any size, glass / mirror / checker / diffuse materials at random."""
    r = np.random.default_rng(seed)
    hs, os_ = rt.HostScene(), O.Scene()
    meshes = []
    for k in range(int(r.integers(1, 5))):
        nt = int(r.integers(200, 6000))
        ext = float(10.0 ** r.uniform(-0.5, 0.7))
        size = float(10.0 ** r.uniform(-2.0, -0.3)) * ext
        c = r.uniform(-ext, ext, (nt, 3))
        e1, e2 = r.uniform(-size, size, (nt, 3)), r.uniform(-size, size, (nt, 3))
        big = r.random(nt) < (0.0 if plain else 0.01)          # plain: no huge triangles and a deep tree -> hardly a fat leaf: the plain 4-wide kernel (k_shadow<1>, the one C5 runs)
        e1[big] *= 20.0
        e2[big] *= 20.0
        centre = r.uniform(-2, 2, 3)
        v = np.stack([c, c + e1, c + e2], axis=1).reshape(-1, 3) + centre
        t = np.arange(3 * nt, dtype=np.uint32).reshape(nt, 3)
        if r.random() < 0.5 and not plain:             # shared vertices: a random re-indexing of a tenth of the corners
            m = r.random(3 * nt) < 0.1
            t = t.reshape(-1)
            t[m] = r.integers(0, 3 * nt, int(m.sum()))
            t = t.reshape(nt, 3)
        meshes.append((v.astype(np.float32), t.astype(np.uint32)))
    spheres = [(r.uniform(-5, 5, 3).astype(np.float32), float(10.0 ** r.uniform(-1.3, 0.3))) for _ in range(int(r.integers(0, 41)))]
    wall = np.array([[-9, -5, -8], [9, -5, -8], [9, 7, -8], [-9, 7, -8]], np.float32)
    floor = np.array([[-9, -5, 7], [9, -5, 7], [9, -5, -8], [-9, -5, -8]], np.float32)
    mesh_depth, top_depth = int(r.integers(2, 23)), int(r.integers(1, 7))
    if plain:
        mesh_depth = 24
        spheres = spheres[:3]
    lights = [(r.uniform(-4, 6, 3).astype(np.float32).tolist(), float(10.0 ** r.uniform(-1.5, 0.8))) for _ in range(int(r.integers(1, 5)))]
    tg = np.random.default_rng(seed + 12345)                   # every third seed: image textures and normal maps (Material.h:86-126) on random materials
    textured = seed % 3 == 0
    images = [tg.integers(0, 256, (int(tg.integers(1, 40)), int(tg.integers(1, 40)), int(tg.integers(3, 5)))).astype(np.uint8) for _ in range(3)] if textured else []
    for s, M in ((hs, rt), (os_, O)):
        tex = [s.add_texture(im) for im in images]

        def mat(k):
            m = M.Material()
            g = np.random.default_rng(7919 * seed + k)
            if textured:
                g2 = np.random.default_rng(104729 * seed + k)
                if g2.random() < 0.5:
                    m.texture = tex[int(g2.integers(0, 3))]
                if g2.random() < 0.4:
                    m.normal_map = tex[int(g2.integers(0, 3))]
                m.material_scale = float(g2.choice([1.0, 0.5, 3.0, 7.3]))
            c = g.uniform(0.1, 1.0, 3)
            m.kd[:] = c.tolist()
            m.ks[:] = g.uniform(0.1, 1.0, 3).tolist()
            m.shininess = float(g.uniform(1.0, 40.0))
            m.type = int(g.integers(0, 4))
            if mirrors:                                   # no glass (two children per hit: the reference's recursion doubles per bounce), many mirrors
                m.type = int(g.choice([0, 2, 2, 2, 3]))
            m.ior = float(g.uniform(1.05, 1.8))
            m.transparency = float(g.uniform(0.1, 0.9)) if m.type == 1 else 0.0
            m.reflectivity = float(g.uniform(0.2, 0.9)) if m.type == 2 else 0.0
            return m
        for lp, rad in lights:
            s.add_light(lp, rad)
        s.add_square(wall, mat(0))
        s.add_square(floor, mat(1))
        for k, (c, rad) in enumerate(spheres):
            s.add_sphere(c, rad, mat(10 + k))
        for k, (v, t) in enumerate(meshes):
            s.add_mesh(v, t, mat(100 + k))
        s.build(top_depth, mesh_depth)
    return hs, os_, lights, meshes, r


def main():
    seed, out = int(sys.argv[1]), sys.argv[2]
    wide = len(sys.argv) > 3 and sys.argv[3] == "wide"            # the second family (build_wide)
    os.makedirs(out, exist_ok=True)
    os.environ["RT_DUMP_SCENE"] = out
    rt = importlib.import_module("ray-tracing_amd")
    hs, os_, lights, meshes, r = (build_wide if wide else build)(seed, rt)
    ctx = rt.Context(0)
    ctx.upload_scene(hs)
    ctx.close()
    oc = O.default_camera(1.0, zoom=0.0, z=-10.0 if wide else -9.0)
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
