"""GPU parity: the HIP path through the C-ABI against the CPU oracle on the same inputs.

Bars (BASELINE.json north_star): hit identities (object ordinal, triangle index) exact -- and here
also t / position / normal bit-exact; radiance max |delta| <= 1/255 on the 8-bit output
(PSNR >= 50 dB for stochastic soft-shadow configs).
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from helpers import psnr

pytestmark = pytest.mark.gpu

SCENES = ["final", "refraction", "reflexion", "reflexion_chamber", "cornell_box", "mesh_gorilla", "mesh_cow", "soup:20000"]
_cache = {}


def scene_pair(rt, ctx, name, mesh_depth=12):
    key = (name, mesh_depth)
    if key not in _cache:
        _cache[key] = (rt.HostScene(name, 3, mesh_depth), O.Scene(name, 3, mesh_depth))
    hs, os_ = _cache[key]
    ctx.upload_scene(hs)
    return hs, os_


def cams(rt, aspect=1.0):
    near = (rt.default_camera(aspect), O.default_camera(aspect))
    far = (rt.default_camera(aspect, zoom=25.0), O.default_camera(aspect, zoom=25.0))
    return [near, far]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_hits_equal(g, o, what):
    assert np.array_equal(g["exists"], o["exists"]), f"{what}: exists differs at {np.flatnonzero(g['exists'] != o['exists'])[:8]}"
    assert np.array_equal(g["obj"], o["obj"]), f"{what}: object ids differ at {np.flatnonzero(g['obj'] != o['obj'])[:8]}"
    assert np.array_equal(g["tri"], o["tri"]), f"{what}: triangle ids differ at {np.flatnonzero(g['tri'] != o['tri'])[:8]}"
    m = o["exists"] == 1
    for f in ("t", "pos", "nrm", "u", "v"):
        assert np.array_equal(bits(g[f][m]), bits(o[f][m])), f"{what}: field {f} not bit-exact"


@pytest.mark.parametrize("jitter", [0, 1])
def test_camera_rays_bit_exact(rt, ctx, jitter):
    for aspect, quat in ((1.0, (0, 0, 0, 1)), (16 / 9, (0.1, -0.2, 0.05, 0.97))):
        gc, oc = rt.default_camera(aspect), O.default_camera(aspect)
        q = np.array(quat, np.float32)
        q /= np.linalg.norm(q)
        gc.quat[:] = q.tolist()
        oc.quat[:] = q.tolist()
        ctx.set_camera(gc)
        gp = rt.make_params(64, 36, spp=3, jitter=jitter, seed=7)
        op = O.make_params(64, 36, spp=3, jitter=jitter, seed=7)
        go, gd = ctx.camera_rays(gp, 5, 3, 40, 20)
        oo, od = O.camera_rays(oc, op, 5, 3, 40, 20)
        assert np.array_equal(bits(go), bits(oo))
        assert np.array_equal(bits(gd), bits(od))


@pytest.mark.parametrize("name", SCENES)
def test_primary_hits_exact(rt, ctx, name):
    hs, os_ = scene_pair(rt, ctx, name)
    for gc, oc in cams(rt):
        op = O.make_params(96, 96, spp=1)
        org, dirs = O.camera_rays(oc, op, 0, 0, 96, 96)
        assert_hits_equal(ctx.intersect(org, dirs), os_.intersect(org, dirs), f"{name} zoom {oc.zoom}")


@pytest.mark.parametrize("name", SCENES)
def test_random_and_axis_rays_exact(rt, ctx, name):
    hs, os_ = scene_pair(rt, ctx, name)
    rng = np.random.default_rng(1234)
    n = 6000
    org = rng.uniform(-9, 9, (n, 3)).astype(np.float32)
    org[:, 1] = rng.uniform(-1.9, 6, n)
    dirs = rng.normal(size=(n, 3)).astype(np.float32)
    # adversarial: axis-aligned directions (1/0 = inf in the slab test), unnormalised, tiny components
    dirs[:600] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 600)] * rng.choice([-1, 1], (600, 1)).astype(np.float32)
    dirs[600:900, 0] = 0.0
    dirs[900:1200] *= 1e-3
    dirs[1200:1500, 1] *= 1e-6
    assert_hits_equal(ctx.intersect(org, dirs), os_.intersect(org, dirs), name)


@pytest.mark.parametrize("name", SCENES)
def test_shadow_queries_exact(rt, ctx, name):
    hs, os_ = scene_pair(rt, ctx, name)
    gc, oc = cams(rt)[1]
    org, dirs = O.camera_rays(oc, O.make_params(64, 64, spp=1), 0, 0, 64, 64)
    h = os_.intersect(org, dirs)
    pts = h["pos"][h["exists"] == 1]
    if len(pts) == 0:
        pytest.skip("no hits")
    rng = np.random.default_rng(5)
    nl = os_.counts()[3]
    lights = np.stack([os_.light(i % nl)[0] for i in range(len(pts))]).astype(np.float32)
    lights2 = lights + rng.normal(scale=3.0, size=lights.shape).astype(np.float32)
    for lp in (lights, lights2):
        g = ctx.occluded(pts, lp)
        o = os_.shadow(pts, lp)
        assert np.array_equal(g, o), f"{name}: {np.count_nonzero(g != o)} of {len(o)} shadow results differ"
    assert 0 < o.sum() or name in ("refraction", "reflexion"), "test should see some occlusion"


def radiance_check(g, o, what, tol_q=1):
    # NaN radiance (e.g. NaN vertex normals next to a degenerate triangle) passes through both paths
    assert np.array_equal(np.isnan(g), np.isnan(o)), f"{what}: NaN pixels differ"
    g, o = np.nan_to_num(g, nan=0.0), np.nan_to_num(o, nan=0.0)
    qg, qo = O.quantize(g), O.quantize(o)
    dq = np.abs(qg - qo)
    rel = np.abs(g - o) / (1e-3 + np.abs(o))
    # main.cpp:253 writes (int)(255 * min(1, v)) and does not clamp below: a radiance of -5e4 (a glass of random parameters under an
    # unnormalised normal) becomes -1.2e7 "levels", of which the last few are the float rounding of the sum (fixed-point here, sequential
    # floats there: DESIGN section 6, deviation 3). One level of 255 is the bar where a level means something, |v| <= 4; beyond, 1e-4 of the pixel's magnitude (device pow / exp differ from glibc's by an ulp, and a specular term of 1e4 carries that).
    big = np.max(np.abs(o), axis=-1, keepdims=True) * np.ones_like(o)       # (a channel can be the small difference of large terms: the pixel's largest channel is the scale)
    far = big > 4.0
    dq = np.where(far, 0, dq)
    assert dq.max() <= tol_q, f"{what}: 8-bit output differs by {dq.max()} levels at {np.argwhere(dq > tol_q)[:4]}"
    relbig = np.abs(g - o) / np.maximum(big, 1.0)
    assert relbig[far].max(initial=0.0) <= 1e-4, f"{what}: out-of-range radiance differs by {relbig[far].max():.3g} of the pixel's magnitude"
    return float(rel.max())


@pytest.mark.parametrize("name", SCENES)
@pytest.mark.parametrize("shadow_mode", [1, 0])
def test_ray_trace_radiance(rt, ctx, name, shadow_mode):
    """Scene::rayTrace per ray (depth 50, 35 light samples): tolerance 1/255 on the 8-bit value."""
    hs, os_ = scene_pair(rt, ctx, name)
    res = 40 if name == "final" else 64
    for gc, oc in cams(rt):
        # the pulled-back view of the default scene sees the glass gorilla: at depth 50 one pixel owns
        # millions of rays, minutes for the single-threaded oracle -> depth 7 there
        depth = 7 if (name == "final" and oc.zoom > 3.0) else 50
        op = O.make_params(res, res, spp=1, shadow_mode=shadow_mode, seed=3, max_depth=depth)
        gp = rt.make_params(res, res, spp=1, shadow_mode=shadow_mode, jitter=0, seed=3, max_depth=depth)
        org, dirs = O.camera_rays(oc, op, 0, 0, res, res)
        pix = np.arange(len(org), dtype=np.uint32)
        g = ctx.trace(org, dirs, gp, pix)
        o = os_.trace(org, dirs, op, pix)
        # float tolerance written out: |delta| <= 1/255 after min(1, .) quantisation (north_star)
        rel = radiance_check(g, o, f"{name} mode {shadow_mode} zoom {oc.zoom}")
        assert np.allclose(np.minimum(g, 4.0), np.minimum(o, 4.0), atol=1.0 / 255.0)
        assert rel < 0.05


@pytest.mark.parametrize("name,w,h,spp,depth", [("final", 64, 64, 2, 50), ("refraction", 160, 90, 4, 5),
                                                 ("mesh_gorilla", 128, 72, 2, 50), ("soup:20000", 96, 54, 4, 5),
                                                 ("reflexion_chamber", 80, 80, 2, 12), ("cornell_box", 96, 96, 2, 50)])
def test_render_frame(rt, ctx, name, w, h, spp, depth):
    hs, os_ = scene_pair(rt, ctx, name)
    aspect = w / h
    gc, oc = rt.default_camera(aspect), O.default_camera(aspect)
    gp = rt.make_params(w, h, spp=spp, max_depth=depth, jitter=1, seed=11)
    op = O.make_params(w, h, spp=spp, max_depth=depth, jitter=1, seed=11)
    rgb, prim, st = ctx.render(gc, gp)
    orgb, oprim, ost, _ = os_.render(oc, op, threads=8)
    oprim_u = oprim.astype(np.int64)
    oprim_u[oprim[..., 0] < 0] = 0xFFFFFFFF
    assert np.array_equal(prim.astype(np.int64), oprim_u), "primary-hit primitive ids differ"
    assert (st.rays_primary, st.rays_secondary) == (ost.rays_primary, ost.rays_secondary)
    assert st.rays_shadow == ost.rays_shadow
    radiance_check(rgb, orgb, name)
    assert psnr(rgb, orgb) >= 50.0
    assert st.nonfinite == 0


def test_tile_sharding_bit_identical(rt, ctx):
    """1 GPU vs the union of the interleaved tile sets of 2 and 4 ranks: same bits."""
    hs, os_ = scene_pair(rt, ctx, "final")
    cam = rt.default_camera(1.0, zoom=25.0)
    p = rt.make_params(96, 96, spp=2, seed=5, max_depth=8)
    full, fprim, _ = ctx.render(cam, p)
    for world in (2, 4):
        acc = np.zeros_like(full)
        aprim = np.full_like(fprim, 0xFFFFFFFF)
        for rank in range(world):
            rgb, prim, _ = ctx.render(cam, p, tiles=rt.interleaved_tiles(96, 96, rank, world))
            acc += rgb
            aprim = np.minimum(aprim, prim)
        assert np.array_equal(bits(acc), bits(full))
        assert np.array_equal(aprim, fprim)
    again, _, _ = ctx.render(cam, p)
    assert np.array_equal(bits(again), bits(full)), "run-to-run determinism"


def test_small_batches_same_frame(rt, ctx, monkeypatch):
    """A tiny wavefront batch (many iterations, deep LIFO stack) gives the same frame."""
    hs, os_ = scene_pair(rt, ctx, "final")
    cam = rt.default_camera(1.0)
    p = rt.make_params(48, 48, spp=2, seed=1)
    full, fprim, st = ctx.render(cam, p)
    monkeypatch.setenv("RT_BATCH", "1024")
    c2 = rt.Context(0)
    try:
        c2.upload_scene(hs)
        rgb, prim, st2 = c2.render(cam, p)
    finally:
        c2.close()
    assert np.array_equal(bits(rgb), bits(full)) and np.array_equal(prim, fprim)
    assert st2.iterations > st.iterations
    assert st2.rays == st.rays


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["final", "mesh_gorilla", "soup:20000"])
def test_shadow_walk_variants_same_frame(rt, ctx, monkeypatch, name):
    """The quantised 4-wide shadow tree (default) and the exact 64-byte nodes give identical frames, with and without
    sub-trees inside fat leaves; a tree whose boxes are not nested switches the quantised nodes off by itself."""
    hs, os_ = scene_pair(rt, ctx, name)
    cam = rt.default_camera(1.0, zoom=25.0) if name == "final" else rt.default_camera(1.0)
    p = rt.make_params(64, 64, spp=2, max_depth=6, seed=3)
    quant, qprim, qst = ctx.render(cam, p)
    monkeypatch.setenv("RT_SHADOW_EXACT_NODES", "1")
    c2 = rt.Context(0)
    try:
        c2.upload_scene(hs)
        exact, eprim, est = c2.render(cam, p)
    finally:
        c2.close()
    monkeypatch.delenv("RT_SHADOW_EXACT_NODES")
    assert np.array_equal(bits(quant), bits(exact)) and np.array_equal(qprim, eprim)
    assert qst.rays_shadow == est.rays_shadow
    # almost no LDS stack: the 4-wide walk spills its pending links to global memory
    monkeypatch.setenv("RT_WIDE_LDS_CAP", "1")
    c4 = rt.Context(0)
    try:
        c4.upload_scene(hs)
        spilled, sprim, sst = c4.render(cam, p)
    finally:
        c4.close()
    monkeypatch.delenv("RT_WIDE_LDS_CAP")
    assert np.array_equal(bits(spilled), bits(quant)) and sst.rays_shadow == qst.rays_shadow
    # fat leaves walked triangle by triangle (no sub-trees), and sub-trees in every leaf of more than 3 triangles
    # ... and top-level rounds of k_shadow that never wait / wait for 48 lanes, top-level tables read from global memory
    for knob, value in (("RT_NO_LEAF_SPLIT", "1"), ("RT_FORCE_LEAF_SPLIT", "1"), ("RT_TOP_MIN", "1"), ("RT_TOP_MIN", "48"),
                        ("RT_TOP_GLOBAL", "1"), ("RT_FRONT", "1"), ("RT_FRONT", "0"),       # ... with and without the front pass
                        ("RT_NO_TOP_SPLIT", "1"),                                        # ... top-level leaves scanned object by object (no sub-trees over their objects)
                        ("RT_NO_FLAT_TOP", "1"),                                         # ... the top-level tree walked node by node instead of the flat pass over its leaves' own boxes
                        ("RT_SEPARATE_HASH", "1"),                                       # ... the shared rounds of the counter RNG in a pass of their own (k_hash_records) instead of inside k_closest
                        ("RT_SPREAD_LANES", "0"), ("RT_SPREAD_LANES", "1048576"),           # ... small batches of bounced rays one state per lane / spread thinner than by default
                        ("RT_NO_TAIL_SWAP", "1"),                                        # ... the children of every iteration copied onto the stack (no buffer swap in the tail of the frame)
                        ("RT_GROUP_CULL", "1"), ("RT_GROUP_CULL", "0"),                    # ... the (hit, light) pairs whose 35 rays provably meet nothing dropped on every chunk / never (k_shadow_group)
                        ("RT_SEPARATE_RAYGEN", "1"),                                     # ... primaries written out by k_raygen instead of made inside k_closest
                        ("RT_WIDE_CLOSEST", "0"), ("RT_WIDE_CLOSEST", "1"),                # ... closest hits over the reference's binary tree only / over the 4-wide quantised one in every launch (default: bounced batches)
                        ("RT_NO_CULL", "1")):                                            # ... every hit shaded and every path followed, however little it carries
        monkeypatch.setenv(knob, value)
        c5 = rt.Context(0)
        try:
            c5.upload_scene(hs)
            other, oprim, ost = c5.render(cam, p)
        finally:
            c5.close()
        monkeypatch.delenv(knob)
        assert np.array_equal(bits(other), bits(quant)) and np.array_equal(oprim, qprim), (knob, value)
        if knob == "RT_NO_CULL":        # (path states that carry nothing -- a glass of transparency 1 reflects with weight 0 -- are followed now, and their hits counted)
            assert ost.rays_shadow >= qst.shadow_queries and ost.rays_secondary >= qst.rays_secondary and ost.rays_shadow_culled == 0
        else:
            assert ost.rays_shadow == qst.rays_shadow and ost.rays_secondary == qst.rays_secondary, (knob, value)
    # shrink one child box of a mesh root below its own children's boxes: not nested any more
    dp = hs.desc
    d = dp.contents
    m = d.meshes[0]
    root = d.mesh_nodes[m.node_first + m.root]
    assert root.left >= 0
    keep = [root.lmax[k] for k in range(3)]
    for k in range(3):
        root.lmax[k] = root.lmin[k] + 0.25 * (keep[k] - root.lmin[k])
    monkeypatch.setenv("RT_DEBUG", "1")
    c3 = rt.Context(0)
    try:
        c3._check(c3.L.rt_upload_scene(c3.h, dp), "rt_upload_scene")
        assert c3.L.rt_uses_quantised_shadow_nodes(c3.h) == 0
        c3.set_camera(cam)
        rgb3, _, st3 = c3.render(cam, p)
        assert np.isfinite(rgb3).all() and st3.rays_shadow > 0
    finally:
        c3.close()
        for k in range(3):
            root.lmax[k] = keep[k]
    assert ctx.L.rt_uses_quantised_shadow_nodes(ctx.h) == 1


def test_drop_in_entry_point_writes_ppm(rt, ctx, tmp_path):
    hs, os_ = scene_pair(rt, ctx, "reflexion")
    gc, oc = rt.default_camera(1.0), O.default_camera(1.0)
    gp = rt.make_params(80, 80, spp=2, seed=2)
    path = str(tmp_path / "rendu.ppm")
    ctx.ray_trace_from_camera(hs, gc, gp, path)
    tok = open(path).read().split()
    assert tok[:4] == ["P3", "80", "80", "255"]
    img = np.array(tok[4:], np.int32).reshape(80, 80, 3)
    orgb, _, _, _ = os_.render(oc, O.make_params(80, 80, spp=2, jitter=1, seed=2), threads=8)
    assert np.abs(img - O.quantize(orgb)).max() <= 1


def test_free_form_scene_and_edge_cases(rt, ctx):
    """Empty scene, single-triangle mesh (root leaf), degenerate triangle, light inside geometry."""
    hs, os_ = rt.HostScene(), O.Scene()
    hs.build(3, 12)
    os_.build(3, 12)
    ctx.upload_scene(hs)
    org = np.zeros((4, 3), np.float32)
    dirs = np.array([[0, 0, -1], [0, 1, 0], [1, 0, 0], [0, -1, 0]], np.float32)
    assert ctx.intersect(org, dirs)["exists"].sum() == 0
    p = rt.make_params(1, 1, spp=1, jitter=0)
    assert np.array_equal(bits(ctx.trace(org, dirs, p)), bits(os_.trace(org, dirs, O.make_params(1, 1, spp=1))))

    hs, os_ = rt.HostScene(), O.Scene()
    tri_v = np.array([[-1, -1, -3], [1, -1, -3], [0, 1, -3]], np.float32)
    deg_v = np.array([[0, 0, -2], [0, 0, -2], [1, 1, -2], [2, 0, -2.5]], np.float32)
    for s, M in ((hs, rt), (os_, O)):
        mat = M.Material()
        mat.kd[:] = [0.8, 0.2, 0.2]
        mat.ks[:] = [0.5, 0.5, 0.5]
        mat.shininess = 8.0
        s.add_light([0.2, 0.3, 2.0], 0.25)
        s.add_mesh(tri_v, np.array([[0, 1, 2]], np.uint32), mat)
        s.add_mesh(deg_v, np.array([[0, 1, 2], [0, 2, 3]], np.uint32), mat)
        s.add_sphere([0.0, 0.0, -1.0], 0.5, mat)
        s.build(3, 12)
    ctx.upload_scene(hs)
    oc = O.default_camera(1.0, zoom=0.0, z=-1.5)
    org, dirs = O.camera_rays(oc, O.make_params(48, 48, spp=1), 0, 0, 48, 48)
    assert_hits_equal(ctx.intersect(org, dirs), os_.intersect(org, dirs), "free-form")
    g = ctx.trace(org, dirs, rt.make_params(48, 48, spp=1, jitter=0))
    o = os_.trace(org, dirs, O.make_params(48, 48, spp=1))
    radiance_check(g, o, "free-form")
    assert np.isnan(o).any(), "the degenerate triangle should poison some normals"


def _random_scene_seeds():
    """11-13 and 102 in the suite; RT_SOAK_SEEDS=a:b adds range(a, b) for a soak run (tools/soak.sh)."""
    extra = os.environ.get("RT_SOAK_SEEDS", "")
    if ":" in extra:
        a, b = extra.split(":")
        return [11, 12, 13, 102] + list(range(int(a), int(b)))
    return [11, 12, 13, 102]      # 102: a mirror mesh with a degenerate triangle under fat leaves -- its bounced rays carry NaNs (a memory fault until round 3)


@pytest.mark.parametrize("seed", _random_scene_seeds())
def test_random_scenes(rt, ctx, seed):
    """Random free-form scenes -- overlapping triangle clouds at several scales and tree depths (single-leaf meshes,
    fat leaves, deep trees), slivers and degenerate triangles, spheres, walls, lights inside geometry: hits, shadow
    queries and radiance against the oracle. This is what exercises the conservative shadow tree on geometry nobody
    tuned it for."""
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
        if k == 0:                                    # slivers and a degenerate triangle
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
        s.add_light(meshes[0][0][0].tolist(), 0.2)       # a light inside the first cloud
        s.add_square(wall, mat(0, 0))
        s.add_square(floor, mat(3, 1))
        for k, (c, rad) in enumerate(spheres):
            s.add_sphere(c, rad, mat(k % 3, 10 + k))
        for k, (v, t) in enumerate(meshes):
            s.add_mesh(v, t, mat(0 if k else 2, 20 + k))
        s.build(top_depth, mesh_depth)
    ctx.upload_scene(hs)
    assert ctx.L.rt_uses_quantised_shadow_nodes(ctx.h) == 1
    n = 6000
    org = r.uniform(-5, 5, (n, 3)).astype(np.float32)
    dirs = r.normal(size=(n, 3)).astype(np.float32)
    dirs[:50, 0] = 0.0                                    # axis-parallel and zero-component directions
    dirs[50:80, 1:] = 0.0
    assert_hits_equal(ctx.intersect(org, dirs), os_.intersect(org, dirs), f"random scene {seed}")
    lights = np.where(r.random((n, 1)) < 0.5, np.array([[0.5, 4.0, 2.0]], np.float32), meshes[0][0][0][None, :]).astype(np.float32)
    pts = org.copy()
    pts[: n // 2] = meshes[1][0][r.integers(0, len(meshes[1][0]), n // 2)] + r.normal(size=(n // 2, 3)).astype(np.float32) * 0.05
    assert np.array_equal(ctx.occluded(pts, lights), os_.shadow(pts, lights))
    oc = O.default_camera(1.0, zoom=0.0, z=-9.0)
    co, cd = O.camera_rays(oc, O.make_params(56, 56, spp=1), 0, 0, 56, 56)
    for shadow_mode in (0, 1):
        g = ctx.trace(co, cd, rt.make_params(56, 56, spp=1, jitter=0, max_depth=6, shadow_mode=shadow_mode, seed=seed))
        o = os_.trace(co, cd, O.make_params(56, 56, spp=1, max_depth=6, shadow_mode=shadow_mode, seed=seed))
        radiance_check(g, o, f"random scene {seed} mode {shadow_mode}")


def _frame_seeds():
    extra = os.environ.get("RT_SOAK_FRAME_SEEDS", "")
    if ":" in extra:
        a, b = extra.split(":")
        return [21, 102] + list(range(int(a), int(b)))
    return [21, 102]


@pytest.mark.parametrize("seed", _frame_seeds())
def test_random_scene_frames(rt, ctx, monkeypatch, seed):
    """The frame path (queued tiles, primaries made inside k_closest, jitter, several samples per pixel, the front pass, group culling forced
    on a second context) on the random scenes of test_random_scenes: primary ids and ray counts exact, radiance <= 1/255 against the oracle;
    RT_SOAK_FRAME_SEEDS=a:b adds seeds (tools/soak.sh)."""
    sys_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "emu")
    import sys
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    from dump_seed import build
    hs, os_, lights, meshes, r = build(seed, rt)
    ctx.upload_scene(hs)
    asp = (17 + seed % 37) / (9 + seed % 29)
    gc, oc = rt.default_camera(asp, zoom=0.0, z=-9.0), O.default_camera(asp, zoom=0.0, z=-9.0)
    for shadow_mode in (0, 1):
        w, h, spp = 17 + seed % 37, 9 + seed % 29, 1 + seed % 3                  # ragged tiles at both edges, one to three samples
        gp = rt.make_params(w, h, spp=spp, max_depth=6, jitter=1, shadow_mode=shadow_mode, seed=seed)
        op = O.make_params(w, h, spp=spp, max_depth=6, jitter=1, shadow_mode=shadow_mode, seed=seed)
        rgb, prim, st = ctx.render(gc, gp)
        orgb, oprim, ost, _ = os_.render(oc, op, threads=8)
        oprim_u = oprim.astype(np.int64)
        oprim_u[oprim[..., 0] < 0] = 0xFFFFFFFF
        assert np.array_equal(prim.astype(np.int64), oprim_u), f"seed {seed}: primary-hit primitive ids differ"
        assert (st.rays_primary, st.rays_secondary, st.rays_shadow) == (ost.rays_primary, ost.rays_secondary, ost.rays_shadow)
        assert np.array_equal(np.isinf(rgb), np.isinf(orgb))
        fin = np.isfinite(orgb) | np.isnan(orgb)
        radiance_check(np.where(fin, rgb, 0.0), np.where(fin, orgb, 0.0), f"random frame {seed} mode {shadow_mode}")
        if shadow_mode == 0:
            monkeypatch.setenv("RT_GROUP_CULL", "1")
            monkeypatch.setenv("RT_BATCH", "1024")           # ... and in batches of 1024 states: several iterations, the LIFO stack in use
            c2 = rt.Context(0)
            try:
                c2.upload_scene(hs)
                rgb2, prim2, st2 = c2.render(gc, gp)
            finally:
                c2.close()
            monkeypatch.delenv("RT_GROUP_CULL")
            monkeypatch.delenv("RT_BATCH")
            assert np.array_equal(bits(rgb2), bits(rgb)) and np.array_equal(prim2, prim) and st2.rays_shadow == st.rays_shadow
            # the same frame from a context of three sub-contexts (tiles dealt inside the library, gathered on the first)
            c3 = rt.Context(devices=[0, 0, 0])
            try:
                c3.upload_scene(hs)
                rgb3, prim3, st3 = c3.render(gc, gp)
            finally:
                c3.close()
            assert np.array_equal(bits(rgb3), bits(rgb)) and np.array_equal(prim3, prim) and st3.rays == st.rays
            # ... and its rays through rt_trace and the entry point that allocates nothing (which may refuse a scene whose walks need more
            # spill entries than its scratch holds -- then with RT_ERR_LIMIT, not with a wrong answer)
            ctx.set_camera(gc)
            org, dirs = ctx.camera_rays(gp, 0, 0, w, h)
            pixel = (np.arange(len(org), dtype=np.uint32) // spp).astype(np.uint32)
            sample = (np.arange(len(org), dtype=np.uint32) % spp).astype(np.uint32)
            want = ctx.trace(org, dirs, gp, pixel, sample)
            try:
                got = ctx.trace(org, dirs, gp, pixel, sample, static=True)
            except rt.RtError as e:
                assert "scratch" in str(e) or "limit" in str(e).lower() or "more than" in str(e), str(e)
            else:
                assert np.array_equal(bits(got), bits(want)), f"seed {seed}: rt_trace_static differs from rt_trace"


def _wide_seeds():
    extra = os.environ.get("RT_SOAK_WIDE_SEEDS", "")
    if ":" in extra:
        a, b = extra.split(":")
        return [31, 32, 33] + list(range(int(a), int(b)))
    return [31, 32, 33]          # 33: with random textures and normal maps


@pytest.mark.parametrize("seed", _wide_seeds())
def test_random_scenes_of_the_second_family(rt, ctx, seed):
    """tools/emu/dump_seed.py build_wide: soups of thousands of small triangles with huge ones mixed in, up to 40 spheres, one to four lights of
    any size, every material kind, tree depths 2..22 and top-level depths 1..6 -- the kernel variants the first family rarely selects (plain
    4-wide walk, batched top-level rounds, sub-trees over objects with the flat pass). Hits bit-exact, shadow queries exact, a small frame
    within 1/255 with exact ray counts; RT_SOAK_WIDE_SEEDS=a:b adds seeds."""
    import sys
    sys_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "emu")
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    from dump_seed import build_wide
    hs, os_, lights, meshes, r = build_wide(seed, rt, plain=seed % 4 == 1)      # every fourth: deep trees without fat leaves (the plain 4-wide kernel of C5)
    ctx.upload_scene(hs)
    n = 3000
    org = r.uniform(-6, 6, (n, 3)).astype(np.float32)
    dirs = r.normal(size=(n, 3)).astype(np.float32)
    dirs[:30, 0] = 0.0
    dirs[30:50, 1:] = 0.0
    assert_hits_equal(ctx.intersect(org, dirs), os_.intersect(org, dirs), f"wide scene {seed}")
    lp = np.asarray([lights[int(i)][0] for i in r.integers(0, len(lights), n)], np.float32)
    pts = org.copy()
    v0 = meshes[0][0]
    pts[: n // 2] = v0[r.integers(0, len(v0), n // 2)] + r.normal(size=(n // 2, 3)).astype(np.float32) * 0.03
    assert np.array_equal(ctx.occluded(pts, lp), os_.shadow(pts, lp))
    gc, oc = rt.default_camera(1.0, zoom=0.0, z=-10.0), O.default_camera(1.0, zoom=0.0, z=-10.0)
    for shadow_mode in (0, 1):
        gp = rt.make_params(32, 32, spp=1, max_depth=5, jitter=1, shadow_mode=shadow_mode, seed=seed)
        op = O.make_params(32, 32, spp=1, max_depth=5, jitter=1, shadow_mode=shadow_mode, seed=seed)
        rgb, prim, st = ctx.render(gc, gp)
        orgb, oprim, ost, _ = os_.render(oc, op, threads=8)
        oprim_u = oprim.astype(np.int64)
        oprim_u[oprim[..., 0] < 0] = 0xFFFFFFFF
        assert np.array_equal(prim.astype(np.int64), oprim_u), f"wide scene {seed}: primary-hit primitive ids differ"
        assert (st.rays_primary, st.rays_secondary, st.rays_shadow + st.rays_shadow_culled) == (ost.rays_primary, ost.rays_secondary, ost.rays_shadow)
        # deviation 3 of DESIGN section 6: the fixed-point accumulators report a channel that reaches 6e7 (a specular term under an unnormalised
        # normal out of a random normal map) as infinite, the reference's float sum carries it on
        huge = np.abs(np.nan_to_num(orgb, nan=0.0, posinf=0.0, neginf=0.0)) >= 5.0e7 / max(1, gp.spp)
        assert np.array_equal(np.isinf(rgb) & ~huge, np.isinf(orgb) & ~huge)
        fin = (np.isfinite(orgb) | np.isnan(orgb)) & ~huge & ~np.isinf(rgb)
        radiance_check(np.where(fin, rgb, 0.0), np.where(fin, orgb, 0.0), f"wide scene {seed} mode {shadow_mode}")


def _grid_seeds():
    extra = os.environ.get("RT_SOAK_GRID_SEEDS", "")
    if ":" in extra:
        a, b = extra.split(":")
        return [41, 42, 43] + list(range(int(a), int(b)))
    return [41, 42, 43]


@pytest.mark.parametrize("seed", _grid_seeds())
def test_rays_through_shared_vertices_and_edges(rt, ctx, seed):
    """Height fields and flat grids (every inner vertex shared by six triangles, every edge by two), one to three of them stacked, and rays aimed
    EXACTLY at vertices, edge midpoints and along grid lines: several triangles of several leaves answer with the very same t, and the
    reference keeps the first it meets (Mesh.cpp:114, strict <; near child first, AccelerationStruct.cpp:273). Hit identities and bits
    against the oracle for the 4-wide walk (rt_intersect) at tree depths 1..16; shadow queries from the vertices; RT_SOAK_GRID_SEEDS=a:b."""
    r = np.random.default_rng(seed)
    hs, os_ = rt.HostScene(), O.Scene()
    n = int(r.integers(4, 28))
    flat = bool(r.integers(0, 2))
    layers = int(r.integers(1, 4))
    step = float(2.0 ** r.integers(-4, 1))                       # a power of two: grid coordinates are exact
    xs = (np.arange(n + 1) - n // 2) * step
    gx, gy = np.meshgrid(xs, xs, indexing="ij")
    meshes = []
    for k in range(layers):
        z = np.zeros_like(gx) if flat else np.round(r.normal(size=gx.shape) * 4) * (step / 4)
        v = np.stack([gx, gy, z + k * (0.0 if r.random() < 0.5 else step)], axis=-1).reshape(-1, 3).astype(np.float32)
        idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
        a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel()
        t = np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)]).astype(np.uint32)
        t = t[r.permutation(len(t))]
        meshes.append((v, t))
    mesh_depth = int(r.integers(1, 17))
    for s, M in ((hs, rt), (os_, O)):
        m = M.Material()
        m.kd[:] = [0.7, 0.6, 0.5]
        m.ks[:] = [0.3, 0.3, 0.3]
        m.shininess = 8.0
        s.add_light([0.3, 0.2, 5.0], 0.5)
        for v, t in meshes:
            s.add_mesh(v, t, m)
        s.build(2, mesh_depth)
    ctx.upload_scene(hs)
    v = meshes[0][0]
    tri = meshes[0][1]
    targets = [v[r.integers(0, len(v), 1500)],                                                    # vertices
               0.5 * (v[tri[r.integers(0, len(tri), 1500), 0]] + v[tri[r.integers(0, len(tri), 1500), 1]]),   # midpoints of vertex pairs (edges among them)
               v[r.integers(0, len(v), 500)] + np.array([step / 2, 0, 0], np.float32)]            # points on grid lines
    tg = np.concatenate(targets).astype(np.float32)
    org = np.empty_like(tg)
    k = np.arange(len(tg)) % 4
    org[:] = np.array([0.0, 0.0, 8.0], np.float32)                                                # a pin-hole above the grid
    org[k == 1] = tg[k == 1] + np.array([0.0, 0.0, 4.0], np.float32)                              # straight down: axis-parallel
    org[k == 2] = tg[k == 2] + np.array([2.0, 0.0, 2.0], np.float32)                              # in a grid plane
    org[k == 3] = tg[k == 3] + np.array([1.0, 1.0, 1.0], np.float32)                              # along a cell diagonal
    dirs = (tg - org).astype(np.float32)
    assert_hits_equal(ctx.intersect(org, dirs), os_.intersect(org, dirs), f"grid scene {seed} (n {n}, flat {flat}, layers {layers}, depth {mesh_depth})")
    lp = np.tile(np.array([[0.3, 0.2, 5.0]], np.float32), (len(tg), 1))
    assert np.array_equal(ctx.occluded(tg, lp), os_.shadow(tg, lp))
    g = ctx.trace(org[:1024], dirs[:1024], rt.make_params(1024, 1, spp=1, jitter=0, max_depth=3, shadow_mode=1, seed=seed))
    o = os_.trace(org[:1024], dirs[:1024], O.make_params(1024, 1, spp=1, max_depth=3, shadow_mode=1, seed=seed))
    radiance_check(g, o, f"grid scene {seed}")


def _hall_seeds():
    extra = os.environ.get("RT_SOAK_HALL_SEEDS", "")
    if ":" in extra:
        a, b = extra.split(":")
        return [51, 52] + list(range(int(a), int(b)))
    return [51, 52]


@pytest.mark.parametrize("seed", _hall_seeds())
def test_random_halls_of_mirrors_at_depth_50(rt, ctx, monkeypatch, seed):
    """The second family without glass and with three mirrors in five materials, at the reference's depth limit (Scene.h:284): paths that
    bounce fifty times between mirror spheres, walls and soups -- the LIFO stack, the tail of small batches, the spread of bounced rays and
    contribution culling (states whose whole subtree cannot move an accumulator) on scenes nobody tuned them for. Radiance against the oracle
    with culling, exact ray counts on an RT_NO_CULL context with the same bits; RT_SOAK_HALL_SEEDS=a:b adds seeds."""
    import sys
    sys_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "emu")
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    from dump_seed import build_wide
    hs, os_, lights, meshes, r = build_wide(seed, rt, mirrors=True)
    ctx.upload_scene(hs)
    gc, oc = rt.default_camera(1.0, zoom=0.0, z=-10.0), O.default_camera(1.0, zoom=0.0, z=-10.0)
    gp = rt.make_params(24, 24, spp=1, max_depth=50, jitter=1, shadow_mode=0, seed=seed)
    op = O.make_params(24, 24, spp=1, max_depth=50, jitter=1, shadow_mode=0, seed=seed)
    rgb, prim, st = ctx.render(gc, gp)
    orgb, oprim, ost, _ = os_.render(oc, op, threads=8)
    huge = np.abs(np.nan_to_num(orgb, nan=0.0, posinf=0.0, neginf=0.0)) >= 5.0e7
    fin = (np.isfinite(orgb) | np.isnan(orgb)) & ~huge & ~np.isinf(rgb)
    radiance_check(np.where(fin, rgb, 0.0), np.where(fin, orgb, 0.0), f"hall {seed}")
    monkeypatch.setenv("RT_NO_CULL", "1")
    c2 = rt.Context(0)
    try:
        c2.upload_scene(hs)
        rgb2, prim2, st2 = c2.render(gc, gp)
    finally:
        c2.close()
    monkeypatch.delenv("RT_NO_CULL")
    assert np.array_equal(bits(rgb2), bits(rgb)) and np.array_equal(prim2, prim), f"hall {seed}: culling changed the frame"
    assert (st2.rays_primary, st2.rays_secondary, st2.rays_shadow) == (ost.rays_primary, ost.rays_secondary, ost.rays_shadow)
    assert st.rays_secondary <= st2.rays_secondary and st.rays_shadow + st.rays_shadow_culled <= st2.rays_shadow


@pytest.mark.parametrize("name", ["mesh_gorilla", "final", "soup:20000"])
def test_rays_with_nan_and_infinite_components(rt, ctx, name):
    """Rays no camera makes but a bounce off a degenerate triangle's normal does (Mesh.cpp:130 normalises a zero vector), and the seams
    accept: NaN and infinite origins and directions, zero directions, origins far outside the scene. A NaN plane time is dropped by the
    conservative box test, so such a ray "hits" every slot of a 4-wide node, the unused ones included; the walks must not follow those
    (a memory fault until round 3). Hits, shadow answers and radiance as the oracle's, which walks the reference's tree with the same rays."""
    hs, os_ = scene_pair(rt, ctx, name, 12 if not name.startswith("soup") else 16)
    r = np.random.default_rng(5)
    n = 1536
    org = r.uniform(-3, 3, (n, 3)).astype(np.float32)
    dirs = r.normal(size=(n, 3)).astype(np.float32)
    k = np.arange(n) % 8
    org[k == 0] = np.nan
    dirs[k == 0] = np.nan
    dirs[k == 1, 0] = np.nan
    org[k == 2, 1] = np.inf
    org[k == 3] = np.array([1e30, -1e30, 1e28], np.float32)
    dirs[k == 4] = 0.0
    org[k == 5, 2] = np.nan
    dirs[k == 6, 1] = -np.inf
    assert_hits_equal(ctx.intersect(org, dirs), os_.intersect(org, dirs), f"{name}: odd rays")
    lights = r.uniform(-4, 6, (n, 3)).astype(np.float32)
    pts = org.copy()
    assert np.array_equal(ctx.occluded(pts, lights), os_.shadow(pts, lights))
    g = ctx.trace(org, dirs, rt.make_params(n, 1, spp=1, jitter=0, max_depth=4, shadow_mode=0, seed=3))
    o = os_.trace(org, dirs, O.make_params(n, 1, spp=1, max_depth=4, shadow_mode=0, seed=3))
    assert np.array_equal(np.isnan(g), np.isnan(o)) and np.array_equal(np.isinf(g), np.isinf(o))
    m = np.isfinite(o)
    assert np.max(np.abs(g[m] - o[m]), initial=0.0) <= 1.0 / 255.0


@pytest.mark.parametrize("mesh_depth", [3, 12])
def test_equal_t_in_different_leaves(rt, ctx, monkeypatch, mesh_depth):
    """Two overlapping triangle grids in the plane z = -3 (normals exactly +-z, so every triangle of the mesh gives the very
    same t for a ray): each ray that hits the mesh hits two or more triangles at one t, mostly in different leaves, and the
    reference keeps the one whose leaf it reaches first (Mesh.cpp:114 strict <, BVH_Node::intersect's near-child-first
    order). The closest-hit walk over the 4-wide tree visits leaves in its own order and has to name the same triangle."""
    n = 24
    xs = np.arange(n + 1, dtype=np.float32) * 0.25 - 3.0
    gx, gy = np.meshgrid(xs, xs, indexing="xy")
    base = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, -3.0, np.float32)], axis=1)
    shifted = base + np.array([0.125, 0.0625, 0.0], np.float32)
    verts = np.concatenate([base, shifted]).astype(np.float32)
    tris = []
    for off in (0, base.shape[0]):
        for j in range(n):
            for i in range(n):
                a = off + j * (n + 1) + i
                b, c, d = a + 1, a + n + 1, a + n + 2
                tris.append([a, b, d] if (i + j) % 2 else [a, d, b])      # both windings: the normal's sign does not change t
                tris.append([a, d, c])
    tris = np.array(tris, np.uint32)
    hs, os_ = rt.HostScene(), O.Scene()
    for s, M in ((hs, rt), (os_, O)):
        m = M.Material()
        m.kd[:] = [0.7, 0.7, 0.7]
        s.add_light([0.0, 0.0, 4.0], 0.3)
        s.add_mesh(verts, tris, m)
        s.build(1, mesh_depth)
    r = np.random.default_rng(5 + mesh_depth)
    k = 30000
    aim = np.stack([r.uniform(-3.2, 3.3, k), r.uniform(-3.2, 3.3, k), np.full(k, -3.0)], axis=1)
    aim[:2000, :2] = np.round(aim[:2000, :2] * 4.0) / 4.0                    # through grid vertices and along edges
    aim[2000:4000, 0] = np.round(aim[2000:4000, 0] * 4.0) / 4.0
    org = np.stack([r.uniform(-6, 6, k), r.uniform(-6, 6, k), r.uniform(-1.0, 5.0, k)], axis=1)
    dirs = (aim - org).astype(np.float32)
    org = org.astype(np.float32)
    oh = os_.intersect(org, dirs)
    assert oh["exists"].mean() > 0.8
    ctx.upload_scene(hs)
    assert_hits_equal(ctx.intersect(org, dirs), oh, "coplanar grids")
    monkeypatch.setenv("RT_WIDE_CLOSEST", "0")
    c2 = rt.Context(0)
    try:
        c2.upload_scene(hs)
        assert_hits_equal(c2.intersect(org, dirs), oh, "coplanar grids, binary walk")
    finally:
        c2.close()


@pytest.mark.parametrize("name", ["refraction", "reflexion"])
def test_direct_shadow_kernel_equals_the_persistent_one(rt, ctx, monkeypatch, name):
    """Scenes without meshes take k_shadow_direct (one thread per shadow ray, no ray buffer); the persistent kernel gives
    the same frame, ray counts included."""
    hs, os_ = scene_pair(rt, ctx, name)
    assert hs.desc.contents.n_meshes == 0
    cam = rt.default_camera(1.0)
    for shadow_mode in (0, 1):
        p = rt.make_params(96, 96, spp=3, max_depth=6, seed=5, shadow_mode=shadow_mode)
        direct, dprim, dst = ctx.render(cam, p)
        monkeypatch.setenv("RT_NO_DIRECT_SHADOW", "1")
        try:
            persistent, pprim, pst = ctx.render(cam, p)
        finally:
            monkeypatch.delenv("RT_NO_DIRECT_SHADOW")
        assert np.array_equal(bits(direct), bits(persistent)) and np.array_equal(dprim, pprim)
        assert (dst.rays_primary, dst.rays_secondary, dst.rays_shadow) == (pst.rays_primary, pst.rays_secondary, pst.rays_shadow)
        assert dst.rays_shadow > 0


@pytest.mark.parametrize("mesh_depth", [2, 14])
def test_sliver_clouds_and_distance_pruning(rt, ctx, mesh_depth):
    """Needle triangles (height / length from 1e-3 down to 1e-8, any orientation, away from the origin): their stored
    area is mostly rounding noise, so Triangle.h:101-107 accepts points well outside them -- or nowhere. The upload
    bounds how far (tri_excursion in rtgpu.hip) and the pruning walks rely on that bound; rays aimed along the
    needles' extended axes, lights just beyond the aim point (so that the distance bound is tight)."""
    r = np.random.default_rng(77 + mesh_depth)
    ns = 400
    p = r.uniform(-3, 3, (ns, 3))
    e = r.normal(size=(ns, 3)); e /= np.linalg.norm(e, axis=1, keepdims=True)
    f = np.cross(e, r.normal(size=(ns, 3))); f /= np.linalg.norm(f, axis=1, keepdims=True)
    L = 10.0 ** r.uniform(-1.0, 0.3, (ns, 1))
    ratio = 10.0 ** r.uniform(-8, -3, (ns, 1))
    v = np.empty((ns, 3, 3))
    v[:, 0] = p
    v[:, 1] = p + e * L
    v[:, 2] = p + e * L * r.uniform(0.1, 0.9, (ns, 1)) + f * L * ratio
    verts = v.reshape(-1, 3).astype(np.float32)
    tris = np.arange(3 * ns, dtype=np.uint32).reshape(ns, 3)
    hs, os_ = rt.HostScene(), O.Scene()
    for s, M in ((hs, rt), (os_, O)):
        m = M.Material()
        m.kd[:] = [0.8, 0.6, 0.4]
        s.add_light([0.0, 6.0, 0.0], 0.3)
        s.add_mesh(verts, tris, m)
        s.build(1, mesh_depth)
    ctx.upload_scene(hs)
    assert ctx.L.rt_uses_quantised_shadow_nodes(ctx.h) == 1
    n = 20000
    k = r.integers(0, ns, n)
    tau = r.uniform(-2.0, 3.0, (n, 1))
    aim = p[k] + e[k] * L[k] * tau + f[k] * L[k] * ratio[k] * r.uniform(-2, 2, (n, 1))
    nrm = np.cross(e[k], f[k])
    tilt = nrm + 0.3 * r.normal(size=(n, 3)) * r.integers(0, 2, (n, 1))
    tilt /= np.linalg.norm(tilt, axis=1, keepdims=True)
    dist = r.uniform(0.5, 6.0, (n, 1))
    org = (aim + tilt * dist).astype(np.float32)
    dirs = (-tilt).astype(np.float32)
    assert_hits_equal(ctx.intersect(org, dirs), os_.intersect(org, dirs), "needles")
    lights = (aim - tilt * 10.0 ** r.uniform(-4, -1, (n, 1))).astype(np.float32)      # just beyond the aim point
    lights[::3] = (aim + tilt * 10.0 ** r.uniform(-4, -1, (n, 1)))[::3].astype(np.float32)   # or just before it
    g, o = ctx.occluded(org, lights), os_.shadow(org, lights)
    assert np.array_equal(g, o)
    assert 0 < int(o.sum()) < n
    for shadow_mode in (0, 1):                  # the persistent shadow kernel on the same rays
        g = ctx.trace(org[:4096], dirs[:4096], rt.make_params(64, 64, spp=1, jitter=0, max_depth=2, shadow_mode=shadow_mode, seed=3))
        o = os_.trace(org[:4096], dirs[:4096], O.make_params(64, 64, spp=1, max_depth=2, shadow_mode=shadow_mode, seed=3))
        radiance_check(g, o, f"needles mode {shadow_mode}")


def test_image_textures_and_normal_maps(rt, ctx):
    """Material::getColorAt / getNormalAt / applyTBN (Material.h:86-126, Scene.h:184-202) on every primitive kind and
    material type: textured diffuse / mirror / glass spheres (u,v from the ray direction), a floor with texture, normal
    map and materialScale 3, a textured + normal-mapped mesh, a 4-channel image."""
    s4 = O.load_ppm(os.path.join(O.IMAGE_DIR, "sphereTextures", "s4.ppm"))
    n3 = O.load_ppm(os.path.join(O.IMAGE_DIR, "normalMaps", "n3.ppm"))
    rgba = np.random.default_rng(5).integers(0, 256, (29, 37, 4)).astype(np.uint8)
    hs, os_ = rt.HostScene(), O.Scene()
    quad = np.array([[-4, -1.5, 2], [4, -1.5, 2], [4, -1.5, -6], [-4, -1.5, -6]], np.float32)
    mv = np.array([[-3, -1.5, -5], [3, -1.5, -5], [3, 2.5, -5.5], [-3, 2.5, -5.5]], np.float32)
    mt = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    for s, M in ((hs, rt), (os_, O)):
        t_s4, t_n3, t_rgba = s.add_texture(s4), s.add_texture(n3), s.add_texture(rgba)

        def mat(**kw):
            m = M.Material()
            m.kd[:] = kw.get("kd", [0.7, 0.7, 0.7])
            m.ks[:] = kw.get("ks", [0.4, 0.4, 0.4])
            m.shininess = 12.0
            m.type = kw.get("type", 0)
            m.ior = kw.get("ior", 1.0)
            m.transparency = kw.get("transparency", 0.0)
            m.reflectivity = kw.get("reflectivity", 0.0)
            m.texture = kw.get("texture", 0)
            m.normal_map = kw.get("normal_map", 0)
            m.material_scale = kw.get("scale", 1.0)
            return m
        s.add_light([1.5, 3.0, 1.0], 0.3)
        s.add_light([-2.0, 2.0, 0.5], 0.0)
        s.add_square(quad, mat(texture=t_s4, normal_map=t_n3, scale=3.0))
        s.add_sphere([-1.6, -0.6, -2.5], 0.9, mat(texture=t_s4))
        s.add_sphere([0.3, -0.7, -3.2], 0.8, mat(type=2, reflectivity=0.8, texture=t_rgba))
        s.add_sphere([1.7, -0.5, -2.2], 0.7, mat(type=1, ior=1.3, transparency=0.6, texture=t_s4, scale=2.0))
        s.add_sphere([0.0, 1.2, -3.0], 0.5, mat(type=3, normal_map=t_n3))
        s.add_mesh(mv, mt, mat(texture=t_rgba, normal_map=t_n3, scale=5.0))
        s.build(3, 12)
    ctx.upload_scene(hs)
    oc = O.default_camera(1.0, zoom=0.0, z=-1.0)
    org, dirs = O.camera_rays(oc, O.make_params(72, 72, spp=1), 0, 0, 72, 72)
    gh, oh = ctx.intersect(org, dirs), os_.intersect(org, dirs)
    assert_hits_equal(gh, oh, "textured")
    kinds = set(oh["obj"][oh["exists"] == 1].tolist())
    assert {0, 1, 2, 3, 4, 5} <= kinds, f"every object should be seen, got {kinds}"
    for shadow_mode in (1, 0):
        g = ctx.trace(org, dirs, rt.make_params(72, 72, spp=1, jitter=0, max_depth=8, shadow_mode=shadow_mode))
        o = os_.trace(org, dirs, O.make_params(72, 72, spp=1, max_depth=8, shadow_mode=shadow_mode))
        radiance_check(g, o, f"textured mode {shadow_mode}")
    assert np.unique(O.quantize(o).reshape(-1, 3), axis=0).shape[0] > 500, "the textures should show"
    # a material that points past the texture table is refused at upload
    dp = hs.desc
    dp.contents.materials[0].texture = 9
    c2 = rt.Context(0)
    try:
        with pytest.raises(rt.RtError, match="texture"):
            c2._check(c2.L.rt_upload_scene(c2.h, dp), "rt_upload_scene")
    finally:
        c2.close()


def test_call_order_and_limits(rt):
    c = rt.Context(0)
    try:
        with pytest.raises(rt.RtError):
            c.begin_frame(rt.make_params(8, 8))          # no scene
        hs = rt.HostScene("reflexion")
        c.upload_scene(hs)
        with pytest.raises(rt.RtError):
            c.render_tile(0, 0, 8, 8)                    # no frame
        c.begin_frame(rt.make_params(8, 8))
        with pytest.raises(rt.RtError):
            c.render_tile(0, 0, 8, 8)                    # no camera
        c.set_camera(rt.default_camera())
        with pytest.raises(rt.RtError):
            c.render_tile(4, 4, 8, 8)                    # outside the image
        with pytest.raises(rt.RtError):
            c.begin_frame(rt.make_params(8, 8, max_depth=1000))
        c.render_tile(0, 0, 0, 0)                        # empty tile is a no-op
        rgb, prim, st = c.download()
        assert st.rays == 0 and not rgb.any()
        # a tree whose links loop, share a child or go deeper than declared is refused at upload
        hs = rt.HostScene("mesh_cow", 3, 6)
        dp = hs.desc                                     # one flattening; the library reads this copy
        d = dp.contents
        upload = lambda: c._check(c.L.rt_upload_scene(c.h, dp), "rt_upload_scene")
        m = d.meshes[0]
        node = d.mesh_nodes[m.node_first + m.root]
        keep = node.left
        assert keep >= 0
        node.left = m.root                               # the root becomes its own child
        with pytest.raises(rt.RtError, match="cycle"):
            upload()
        node.left = keep
        deep = m.depth
        m.depth = 2
        with pytest.raises(rt.RtError, match="deeper"):
            upload()
        m.depth = deep
        upload()                                         # restored: accepted again
    finally:
        c.close()


def test_contribution_culling_does_not_change_a_bit(rt, ctx, monkeypatch):
    """Hits and path states that cannot reach half a unit of the fixed-point accumulators are not traced (k_closest): the
    depth-50 recursion inside the default scene's glass meshes is cut where it stops mattering. The frame must be the
    one that follows every path to the end, bit for bit, on the view where the recursion explodes."""
    hs, _ = scene_pair(rt, ctx, "final")
    cam = rt.default_camera(1.0, zoom=25.0)
    p = rt.make_params(40, 40, spp=1, max_depth=50, light_samples=35, shadow_mode=0, jitter=1, seed=8)
    culled, cprim, cst = ctx.render(cam, p)
    monkeypatch.setenv("RT_NO_CULL", "1")
    full_ctx = rt.Context(0)
    try:
        full_ctx.upload_scene(hs)
        full, fprim, fst = full_ctx.render(cam, p)
    finally:
        full_ctx.close()
    monkeypatch.delenv("RT_NO_CULL")
    assert np.array_equal(bits(culled), bits(full)) and np.array_equal(cprim, fprim)
    assert fst.rays_shadow_culled == 0 and cst.rays_shadow_culled > 0
    assert cst.rays_secondary < fst.rays_secondary and cst.rays_primary == fst.rays_primary
    assert cst.shadow_queries <= fst.rays_shadow          # (hits below culled path states are not even counted)
    print(f"rays traced with / without culling: {cst.rays} / {fst.rays}")


@pytest.mark.gpu
@pytest.mark.parametrize("name,depth,zoom", [("final", 50, 25.0), ("cornell_box", 50, 3.0), ("mesh_gorilla", 50, 3.0), ("no light", 50, 3.0)])
def test_sample_sharded_contexts_add_up_to_the_one_context_frame(rt, ctx, name, depth, zoom):
    """Sample sharding (rt_begin_frame_samples): three contexts -- three ranks -- render disjoint sample ranges of every
    pixel; their accumulators, added (integers) and their flag words OR-ed, resolve to the BITS of the frame one context renders alone,
    non-finite pixels (cornell box: normal maps) included. The addition is done on the host here; dist.SampleReduce does it with an
    all-reduce (tests/test_multigpu_gloo.py)."""
    import ctypes
    import importlib
    rtd = importlib.import_module("ray-tracing_amd.dist")
    # caller-owned device memory from the HIP runtime the library itself uses (by soname, as tests/test_progressive.py does; torch's
    # bundled runtime must not be initialised beside it in this process)
    hip = ctypes.CDLL("libamdhip64.so.7")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    if name == "no light":
        # a scene without lights: the reference's shadow factor is 1 - 0/0 (Scene.h:211-222), every diffuse hit a NaN -> the flag words matter
        hs = rt.HostScene()
        hs.add_sphere([0.0, 0.0, 0.0], 1.0, rt.make_material(kd=(0.8, 0.2, 0.2), shininess=8.0))
        hs.add_sphere([1.5, 0.5, -1.0], 0.7, rt.make_material(kd=(1, 1, 1), type=2, reflectivity=0.8))
        hs.build(3, 12)
        ctx.upload_scene(hs)
    else:
        hs, _ = scene_pair(rt, ctx, name)
    W, H, SPP = 96, 64, 7
    cam = rt.default_camera(W / H, zoom=zoom)
    p = rt.make_params(W, H, spp=SPP, max_depth=depth, light_samples=35, shadow_mode=0, jitter=1, seed=6)
    want, wprim, wst = ctx.render(cam, p)
    world = 3
    d_acc, d_flags = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d_acc), W * H * 3 * 8) == 0 and hip.hipMalloc(ctypes.byref(d_flags), W * H * 4) == 0
    acc, flags = np.zeros((H, W, 3), np.int64), np.zeros((H, W), np.uint32)
    tot_acc, tot_flags = np.zeros_like(acc), np.zeros_like(flags)
    rays = [0, 0, 0]
    ranks = []
    try:
        for r in range(world):
            c = rt.Context(0)
            ranks.append(c)
            c.upload_scene(hs)
            c.set_camera(cam)
            first, count = rtd.sample_range(r, world, SPP)
            c.begin_frame_samples(p, first, count)
            c.render_tiles(rt.interleaved_tiles(W, H))
            c.export_accumulators(d_acc.value, d_flags.value)
            assert hip.hipMemcpy(acc.ctypes.data, d_acc, acc.nbytes, 2) == 0 and hip.hipMemcpy(flags.ctypes.data, d_flags, flags.nbytes, 2) == 0
            tot_acc += acc
            tot_flags |= flags
            st = c.stats()
            rays = [rays[0] + st.rays_primary, rays[1] + st.rays_secondary, rays[2] + st.shadow_queries]
        assert rays[0] == wst.rays_primary == W * H * SPP
        assert hip.hipMemcpy(d_acc, tot_acc.ctypes.data, tot_acc.nbytes, 1) == 0 and hip.hipMemcpy(d_flags, tot_flags.ctypes.data, tot_flags.nbytes, 1) == 0
        got = []
        for c in ranks:                       # every rank resolves the same frame
            c.import_accumulators(d_acc.value, d_flags.value)
            rgb, prim, _ = c.download()
            got.append((rgb, prim))
    finally:
        for c in ranks:
            c.close()
        hip.hipFree(d_acc)
        hip.hipFree(d_flags)
    for rgb, _ in got:
        assert np.array_equal(rgb.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(got[0][1], wprim)                       # sample 0 is rank 0's: it has the primary-hit ids
    assert (got[1][1] == 0xFFFFFFFF).all()
    if name == "no light":
        assert np.isnan(want).any() and np.isfinite(want).any()
    # misuse is refused
    with pytest.raises(rt.RtError):
        ctx.begin_frame_samples(p, 5, 3)                          # 5 + 3 > 7
    ctx.begin_frame_samples(p, 0, 3)
    with pytest.raises(rt.RtError):
        ctx.continue_frame(1)
