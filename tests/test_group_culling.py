"""Group culling of shadow rays (k_shadow_group, rt_kernels.h): the 35 rays of a (hit, light) pair that provably meet nothing are
not made. The answers must be the reference's: frames bit-identical with the test forced on every chunk (RT_GROUP_CULL=1) and
switched off (=0), the number of answered shadow queries the same, and the forced frame against the oracle. Scenes: the
reference's own (floor + mesh, floor + sphere, the mirror chamber, the default scene with its 84 objects and a light of
radius 10.75, the cornell box with a light inside a room of textured walls) and random ones with lights inside geometry,
spheres cut by the floor, walls behind everything."""
import importlib

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def render_with(rt, monkeypatch, hs, cam, p, value):
    monkeypatch.setenv("RT_GROUP_CULL", value)
    c = rt.Context(0)
    try:
        c.upload_scene(hs)
        return c.render(cam, p)
    finally:
        c.close()
        monkeypatch.delenv("RT_GROUP_CULL")


@pytest.mark.parametrize("name,zoom,depth,expect", [("mesh_gorilla", 3.0, 6, True), ("mesh_cow", 3.0, 6, True), ("refraction", 3.0, 5, True),
                                                    ("reflexion", 3.0, 5, True), ("reflexion_chamber", 3.0, 8, True), ("final", 25.0, 6, False),
                                                    ("cornell_box", 3.0, 6, False), ("stage_blob", 3.0, 6, True)])
def test_forced_on_and_off_give_the_same_frame(rt, monkeypatch, name, zoom, depth, expect):
    hs = rt.HostScene(name, 3, 12)
    cam = rt.default_camera(96 / 64, zoom=zoom)
    for shadow_mode in (0, 1):
        p = rt.make_params(96, 64, spp=2, max_depth=depth, light_samples=35, shadow_mode=shadow_mode, jitter=1, seed=5)
        on, onprim, onst = render_with(rt, monkeypatch, hs, cam, p, "1")
        off, offprim, offst = render_with(rt, monkeypatch, hs, cam, p, "0")
        assert np.array_equal(bits(on), bits(off)) and np.array_equal(onprim, offprim), (name, shadow_mode)
        assert (onst.rays_primary, onst.rays_secondary, onst.rays_shadow) == (offst.rays_primary, offst.rays_secondary, offst.rays_shadow)
        assert offst.rays_shadow_grouped == 0 and onst.rays_shadow_grouped <= onst.rays_shadow
        assert onst.rays == offst.rays - onst.rays_shadow_grouped
        if expect:
            assert onst.rays_shadow_grouped > 0, (name, shadow_mode)
        print(f"{name} mode {shadow_mode}: {onst.rays_shadow_grouped} of {onst.rays_shadow} shadow queries settled by the group test")


@pytest.mark.parametrize("name,zoom", [("mesh_gorilla", 3.0), ("refraction", 3.0)])
def test_forced_on_against_the_oracle(rt, monkeypatch, name, zoom):
    hs, os_ = rt.HostScene(name, 3, 12), O.Scene(name, 3, 12)
    p = rt.make_params(64, 48, spp=2, max_depth=5, light_samples=35, shadow_mode=0, jitter=1, seed=3)
    op = O.make_params(64, 48, spp=2, max_depth=5, light_samples=35, shadow_mode=0, jitter=1, seed=3)
    got, gprim, gst = render_with(rt, monkeypatch, hs, rt.default_camera(64 / 48, zoom=zoom), p, "1")
    want, wprim, wst, _ = os_.render(O.default_camera(64 / 48, zoom=zoom), op, threads=8)
    assert gst.rays_shadow_grouped > 0
    wprim_u = wprim.astype(np.int64)
    wprim_u[wprim[..., 0] < 0] = 0xFFFFFFFF
    assert np.array_equal(gprim.astype(np.int64), wprim_u)
    assert int(np.abs(O.quantize(got) - O.quantize(want)).max()) <= 1
    assert (gst.rays_primary, gst.rays_secondary, gst.rays_shadow) == (wst.rays_primary, wst.rays_secondary, wst.rays_shadow)


@pytest.mark.parametrize("seed", [21, 22, 23, 24])
def test_random_scenes_forced_on_and_off(rt, monkeypatch, seed):
    """Free-form scenes built to put lights, hit points and objects in each other's way: a light inside a triangle cloud, one of
    radius 3 next to the spheres, spheres cut by the floor, a big sphere the camera sits in front of, a wall behind the lights."""
    r = np.random.default_rng(seed)
    hs = rt.HostScene()

    def mat(kind, k):
        m = rt.Material()
        c = np.random.default_rng(1000 * seed + k).uniform(0.2, 1.0, 3)
        m.kd[:] = c.tolist()
        m.ks[:] = c.tolist()
        m.shininess = 10.0
        m.type = kind
        m.ior = 1.3
        m.transparency = 0.5 if kind == 1 else 0.0
        m.reflectivity = 0.7 if kind == 2 else 0.0
        return m

    clouds = []
    for k in range(int(r.integers(1, 4))):
        # clouds of separate, reasonably shaped triangles (a degenerate one has no hit bound, and a scene with such a mesh gets no object
        # boxes at all: tests/test_gpu_parity.py::test_random_scenes covers those)
        nt = int(r.integers(1, 300))
        scale = float(10.0 ** r.uniform(-1.0, 0.0))
        centres = r.uniform(-3, 3, 3) + scale * r.normal(size=(nt, 1, 3))
        e = r.normal(size=(nt, 3, 3))
        e /= np.linalg.norm(e, axis=2, keepdims=True)
        v = (centres + 0.3 * scale * e).reshape(-1, 3).astype(np.float32)
        t = np.arange(3 * nt, dtype=np.uint32).reshape(nt, 3)
        clouds.append((v, t))
    hs.add_light(r.uniform(-2, 2, 3).tolist(), float(r.uniform(0.05, 0.6)))
    hs.add_light(clouds[0][0][0].tolist(), 0.2)                  # inside the first cloud
    if seed % 2:
        hs.add_light([3.0, 1.0, 2.0], 3.0)                       # a light as large as the objects
    hs.add_square(np.array([[-8, -4, -7], [8, -4, -7], [8, 6, -7], [-8, 6, -7]], np.float32), mat(0, 0))
    hs.add_square(np.array([[-8, -2, 6], [8, -2, 6], [8, -2, -7], [-8, -2, -7]], np.float32), mat(3, 1))
    for k in range(int(r.integers(1, 7))):
        c = r.uniform(-4, 4, 3).astype(np.float32)
        if k == 0:
            c[1] = -2.0                                          # cut by the floor
        hs.add_sphere(c, float(r.uniform(0.2, 1.5)), mat(0 if k % 3 == 0 else k % 3, 10 + k))
    hs.add_sphere(np.array([0.0, -40.0, 0.0], np.float32), 37.5, mat(0, 40))        # a big one below everything
    for k, (v, t) in enumerate(clouds):
        hs.add_mesh(v, t, mat(0, 20 + k))
    hs.build(int(r.integers(1, 5)), int(r.integers(2, 16)))
    cam = rt.default_camera(1.0, zoom=0.0, z=-9.0)
    for shadow_mode in (0, 1):
        p = rt.make_params(72, 72, spp=1, max_depth=4, light_samples=35, shadow_mode=shadow_mode, jitter=1, seed=seed)
        on, onprim, onst = render_with(rt, monkeypatch, hs, cam, p, "1")
        off, offprim, offst = render_with(rt, monkeypatch, hs, cam, p, "0")
        assert np.array_equal(bits(on), bits(off)) and np.array_equal(onprim, offprim), (seed, shadow_mode)
        assert onst.rays_shadow == offst.rays_shadow
        assert onst.rays_shadow_grouped > 0, (seed, shadow_mode)          # (the far parts of the floor see the small lights with nothing near)
        print(f"seed {seed} mode {shadow_mode}: {onst.rays_shadow_grouped} of {onst.rays_shadow} settled by the group test")


@pytest.mark.parametrize("name", ["mesh_gorilla", "refraction", "reflexion_chamber", "final", "cornell_box", "stage_blob"])
def test_a_clear_pair_has_no_occluded_ray(rt, name):
    """The predicate itself (rt_group_clear), on pairs no frame would produce: hit points all over the scene's surfaces, lights of
    random size anywhere around them. Wherever it says 'clear', Scene::isInShadow (rt_occluded, the exact query) must be false for
    every ray from the point to the light's sphere -- sampled densely on the sphere, its silhouette as seen from the point included."""
    import zlib
    r = np.random.default_rng(zlib.crc32(name.encode()))
    ctx = rt.Context(0)
    try:
        hs = rt.HostScene(name, 3, 12)
        ctx.upload_scene(hs)
        # hit points: what random rays from random places hit
        n = 20000
        org = r.uniform(-6, 6, (n, 3)).astype(np.float32)
        dirs = r.normal(size=(n, 3)).astype(np.float32)
        hits = ctx.intersect(org, dirs)
        pts = hits["pos"][hits["exists"] != 0].astype(np.float32)
        assert len(pts) > 200
        d = hs.desc.contents
        lights = np.array([[d.lights[i].pos[0], d.lights[i].pos[1], d.lights[i].pos[2], d.lights[i].radius] for i in range(d.n_lights)], np.float32).reshape(-1, 4)
        m = len(pts)
        centres = (pts + r.normal(size=(m, 3)) * r.uniform(0.5, 8.0, (m, 1))).astype(np.float32)
        radii = (10.0 ** r.uniform(-2.5, 0.3, m)).astype(np.float32)
        if len(lights):                                   # half of the pairs go to the scene's own lights
            pick = r.integers(0, len(lights), m)
            own = r.random(m) < 0.5
            centres[own] = lights[pick[own], :3]
            radii[own] = lights[pick[own], 3]
        clear = ctx.group_clear(pts, centres, radii)
        idx = np.flatnonzero(clear)
        print(f"{name}: {len(idx)} of {m} pairs clear")
        assert len(idx) >= 5, name
        idx = idx[:400]
        # rays to the sphere: random points on it, and the silhouette circle as seen from the point (the widest rays)
        k = 96
        u = r.normal(size=(len(idx), k, 3))
        u /= np.linalg.norm(u, axis=2, keepdims=True)
        axis = centres[idx] - pts[idx]
        axis /= np.linalg.norm(axis, axis=1, keepdims=True)
        rim = u[:, : k // 2] - (u[:, : k // 2] * axis[:, None, :]).sum(axis=2, keepdims=True) * axis[:, None, :]
        rim /= np.linalg.norm(rim, axis=2, keepdims=True)
        u[:, : k // 2] = rim
        lp = (centres[idx][:, None, :] + radii[idx][:, None, None] * u).astype(np.float32).reshape(-1, 3)
        pp = np.repeat(pts[idx], k, axis=0)
        occ = ctx.occluded(pp, lp)
        assert not occ.any(), f"{name}: {int(occ.sum())} occluded rays in pairs the group test called clear"
    finally:
        ctx.close()
