"""CPU tests of the product's host side (no GPU calls): the C-ABI library loads and exports every
declared symbol, and the host Scene mirror + BVH builders + flattening agree bit for bit with the oracle."""
import os
import re

import numpy as np
import pytest

import oracle_lib as O
from helpers import flat_tree_preorder

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(rt):
    L = rt.lib()
    declared = set()
    for h in ("rtgpu.h", "rthost.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        c_part = txt.split("/* ------------------------------------------------------------------ C facade */")[-1] if h == "rthost.h" else txt
        declared |= set(re.findall(r"\b(rth?_[a-z_0-9]+)\s*\(", c_part))
    declared -= {"rt_status", "rt_material_type"}
    assert declared, "no declarations parsed"
    for sym in sorted(declared):
        assert hasattr(L, sym), f"{sym} declared in include/ but not exported by librtgpu.so"
    assert declared <= set(rt.EXPORTS) | {"rt_ctx"}


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a, np.float32).view(np.uint32), np.ascontiguousarray(b, np.float32).view(np.uint32))


def compare_scene(hs, os_):
    A = hs.arrays()
    nsq, nsp, nm, nl = os_.counts()
    assert (len(A["squares"]), len(A["spheres"]), len(A["meshes"]), len(A["lights"])) == (nsq, nsp, nm, nl)
    mats = A["materials"]

    def mat_eq(i, om):
        m = mats[i]
        return (list(m.kd) == list(om.kd) and list(m.ks) == list(om.ks) and list(m.ka) == list(om.ka) and m.shininess == om.shininess
                and m.type == om.type and m.ior == om.ior and m.transparency == om.transparency and m.reflectivity == om.reflectivity
                and m.texture == om.texture and m.normal_map == om.normal_map and m.material_scale == om.material_scale)

    for i, t in enumerate(A["textures"]):
        assert np.array_equal(t, os_.texture(i))

    for i, s in enumerate(A["squares"]):
        v, om = os_.square(i)
        assert same_bits(list(s.v0), v[0]) and same_bits(list(s.v1), v[1]) and same_bits(list(s.v3), v[3]) and mat_eq(s.material, om)
    for i, s in enumerate(A["spheres"]):
        c, r, om = os_.sphere(i)
        assert same_bits(list(s.center), c) and same_bits([s.radius], [r]) and mat_eq(s.material, om)
    for i, l in enumerate(A["lights"]):
        p, r, col = os_.light(i)
        assert same_bits(list(l.pos), p) and l.radius == r and same_bits(list(l.color), col)
    lf, cnt, bx, it = flat_tree_preorder(A["top_nodes"], A["top_leaves"], A["top_items"], A["top_root"])
    olf, ocnt, obx, oit = os_.tree(-1)
    assert np.array_equal(lf, olf) and np.array_equal(cnt, ocnt) and np.array_equal(it, oit)
    assert same_bits(bx[1:], obx[1:])
    for mi, m in enumerate(A["meshes"]):
        v, n, tr, om = os_.mesh(mi)
        assert same_bits(A["positions"][m.vert_first:m.vert_first + m.vert_count], v)
        assert same_bits(A["normals"][m.vert_first:m.vert_first + m.vert_count], n)
        assert np.array_equal(A["tris"][m.tri_first:m.tri_first + m.tri_count], tr) and mat_eq(m.material, om)
        nodes = A["mesh_nodes"][m.node_first:m.node_first + m.node_count]
        leaves = A["mesh_leaves"][m.leaf_first:m.leaf_first + m.leaf_count]
        items = A["mesh_items"][m.tri_first:m.tri_first + m.tri_count]
        lf, cnt, bx, it = flat_tree_preorder(nodes, leaves, items, m.root)
        olf, ocnt, obx, oit = os_.tree(mi)
        assert np.array_equal(lf, olf) and np.array_equal(cnt, ocnt) and np.array_equal(it, oit)
        assert same_bits(bx[1:], obx[1:])
        assert sorted(it.tolist()) == list(range(m.tri_count)), "every triangle sits in exactly one leaf"


@pytest.mark.parametrize("name,depth", [("final", 12), ("final", 3), ("refraction", 12), ("reflexion", 12), ("reflexion_chamber", 12), ("cornell_box", 12), ("mesh_gorilla", 12),
                                        ("mesh_cow", 16), ("soup:30000", 12), ("soup:30000", 20), ("stage_blob", 12), ("stage_blob_flat", 12), ("stage_gorilla", 12)])
def test_named_scenes_flatten_like_the_oracle(rt, name, depth):
    compare_scene(rt.HostScene(name, 3, depth), O.Scene(name, 3, depth))


def test_free_form_scene_and_top_depth(rt):
    rng = np.random.default_rng(3)
    hs, os_ = rt.HostScene(), O.Scene()
    for s, M in ((hs, rt), (os_, O)):
        r = np.random.default_rng(3)
        mat = M.Material()
        mat.kd[:] = [0.3, 0.6, 0.9]
        mat.shininess = 4.0
        for _ in range(40):
            s.add_sphere(r.uniform(-10, 10, 3).astype(np.float32), float(r.uniform(0.1, 2)), mat)
        for _ in range(5):
            v = r.uniform(-3, 3, (50, 3)).astype(np.float32)
            t = r.integers(0, 50, (80, 3)).astype(np.uint32)
            s.add_mesh(v, t, mat)
        s.add_light([0, 5, 0], 1.0)
        s.build(6, 9)
    compare_scene(hs, os_)
    assert hs.desc.contents.top_depth <= 6


def test_ppm_images_and_texture_tables(rt):
    """The host's PPM reader, the oracle's and PIL agree on the reference's image data; the chamber scene carries its
    normal map; texture indices are 1-based in the C-ABI (0 = none) and checked."""
    from PIL import Image
    path = os.path.join(O.IMAGE_DIR, "sphereTextures", "s4.ppm")
    ref = np.array(Image.open(path))
    assert ref.shape == (256, 512, 3) and np.array_equal(O.load_ppm(path), ref)
    hs, os_ = rt.HostScene("reflexion_chamber", 3, 12), O.Scene("reflexion_chamber", 3, 12)
    A = hs.arrays()
    nm = np.array(Image.open(os.path.join(O.IMAGE_DIR, "normalMaps", "n3.ppm")))
    assert len(A["textures"]) == 1 and np.array_equal(A["textures"][0], nm) and np.array_equal(os_.texture(0), nm)
    floor = A["materials"][A["squares"][3].material]
    assert floor.normal_map == 1 and floor.texture == 0 and floor.type == 3 and floor.material_scale == 1.0
    assert A["materials"][A["squares"][0].material].material_scale == 4.0
    hs2, os2 = rt.HostScene(), O.Scene()
    for s, M in ((hs2, rt), (os2, O)):
        t = s.add_texture(ref)
        assert t == 1
        mat = M.Material()
        mat.texture = 2                                   # no such image
        with pytest.raises(RuntimeError):
            s.add_sphere([0, 0, 0], 1.0, mat)
        mat.texture = t
        s.add_sphere([0, 0, 0], 1.0, mat)
        s.build(3, 12)
    compare_scene(hs2, os2)
    with pytest.raises((RuntimeError, rt.RtError)):
        hs2.add_texture(ref[:, :, :2])                    # fewer than 3 channels


def test_load_off_matches_the_oracle(rt):
    for f in ("flamingo.off", "gorilla18_fixed.obj.off", "cow.off", "triangle.off", "cube.off"):
        path = os.path.join(O.ASSET_DIR, f)
        v, t = rt.load_off(path)
        import ctypes as C
        nv, nt = C.c_int(), C.c_int()
        assert O.lib().orc_load_off(path.encode(), None, 0, None, 0, C.byref(nv), C.byref(nt)) == 0
        ov = np.zeros((nv.value, 3), np.float32)
        ot = np.zeros((nt.value, 3), np.uint32)
        O.lib().orc_load_off(path.encode(), ov.ctypes.data_as(C.c_void_p), nv.value, ot.ctypes.data_as(C.c_void_p), nt.value, C.byref(nv), C.byref(nt))
        assert same_bits(v, ov) and np.array_equal(t, ot)
    with pytest.raises(rt.RtError):
        rt.load_off("/nonexistent/file.off")


def test_limits_are_reported(rt):
    hs = rt.HostScene()
    with pytest.raises(rt.RtError):
        hs.build(3, 1000)           # tree depth above RT_MAX_TREE_DEPTH
    with pytest.raises(rt.RtError):
        rt.HostScene("no_such_scene")
    with pytest.raises(rt.RtError):
        hs.add_mesh(np.zeros((3, 3), np.float32), np.array([[0, 1, 7]], np.uint32), rt.Material())


def test_counter_rng_is_shared(rt):
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = [int(x) for x in rng.integers(0, 2 ** 32, 7, dtype=np.uint64)]
        code = int(rng.integers(0, 2 ** 62, dtype=np.uint64))
        assert rt.rng(a[0], a[1], a[2], code, a[3], a[4], a[5] & 1) == O.rng(a[0], a[1], a[2], code, a[3], a[4], a[5] & 1)


def test_ppm_writer_matches_main_cpp_format(rt, tmp_path):
    rgb = np.random.default_rng(1).uniform(-0.2, 1.5, (5, 7, 3)).astype(np.float32)
    path = str(tmp_path / "rendu.ppm")
    rt.write_ppm(path, rgb)
    txt = open(path).read()
    assert txt.startswith("P3\n7 5\n255\n") and txt.endswith(" \n")
    vals = np.array(txt.split()[4:], np.int32).reshape(5, 7, 3)
    assert np.array_equal(vals, O.quantize(rgb))


def test_interleaved_tiles_partition_the_image(rt):
    for (w, h, world) in ((3840, 2160, 8), (100, 70, 3), (31, 33, 2)):
        seen = np.zeros((h, w), np.int32)
        per_rank = []
        for r in range(world):
            tiles = rt.interleaved_tiles(w, h, r, world)
            per_rank.append(len(tiles))
            for (x, y, tw, th) in tiles:
                seen[y:y + th, x:x + tw] += 1
        assert np.all(seen == 1)
        assert max(per_rank) - min(per_rank) <= 1


def test_threaded_tree_build_equals_the_sequential_one(rt, monkeypatch):
    """buildBVH on several threads (meshes of 200 000 triangles and more: fragments appended in the order the
    recursion emits them) gives the tree of the one-thread build, bit for bit."""
    monkeypatch.setenv("OMP_NUM_THREADS", "1")
    sa = rt.HostScene("soup:500000", 3, 22)              # (kept alive: the mesh records below are views into the scenes)
    a = sa.arrays()
    monkeypatch.setenv("OMP_NUM_THREADS", "8")
    sb = rt.HostScene("soup:500000", 3, 22)
    b = sb.arrays()
    for k in ("mesh_items", "mesh_nodes", "mesh_leaves", "top_nodes", "top_leaves", "top_items"):
        assert a[k].tobytes() == b[k].tobytes(), k
    assert len(a["mesh_nodes"]) > 200000
    assert (a["meshes"][0].root, a["meshes"][0].depth) == (b["meshes"][0].root, b["meshes"][0].depth)
