"""Mesh trees built on the GPU ("BVH construction on GPU"; rthost.h Settings::tree_mode = 2, lbvh.hip):
a linear BVH -- Morton keys, radix sort, Karras' hierarchy, boxes fitted bottom-up -- in the layout of the host builders.
Like the binned-SAH mode it is NOT a parity mode (another topology reaches grazing triangles differently); the tests
check that the tree is a valid one over all triangles and bound the pixels that change their primary hit."""
import importlib
import time

import numpy as np
import pytest

rt = importlib.import_module("ray-tracing_amd")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["mesh_gorilla", "mesh_cow", "soup:50000", "reflexion_chamber"])
def test_gpu_built_tree_is_a_valid_tree_over_all_triangles(name):
    scene = rt.HostScene(name, 3, 12, tree_mode=2)
    A = scene.arrays()
    m = A["meshes"][0]
    assert 1 <= m.depth <= 40
    items = A["mesh_items"][m.tri_first:m.tri_first + m.tri_count]
    assert sorted(items.tolist()) == list(range(m.tri_count))                   # every triangle in exactly one leaf
    nodes = A["mesh_nodes"][m.node_first:m.node_first + m.node_count]
    leaves = A["mesh_leaves"][m.leaf_first:m.leaf_first + m.leaf_count]
    assert m.node_count == m.tri_count - 1 and m.leaf_count == m.tri_count and (leaves["count"] == 1).all()
    pos, tris = A["positions"], A["tris"]
    stack = [(int(m.root), None, None, 0)]
    seen = deepest = 0
    while stack:
        link, mn, mx, level = stack.pop()
        seen += 1
        if link < 0:
            lf = leaves[-1 - link]
            v = pos[tris[m.tri_first + items[lf["first"]]] + m.vert_first]
            assert (v.min(axis=0) == mn).all() and (v.max(axis=0) == mx).all()   # a leaf's box is its triangle's, exactly
            deepest = max(deepest, level)
            continue
        n = nodes[link]
        for side, child in (("l", "left"), ("r", "right")):
            cmn, cmx = n[side + "min"], n[side + "max"]
            assert (cmn <= cmx).all()
            if mn is not None:
                assert (cmn >= mn).all() and (cmx <= mx).all()                   # nested
            stack.append((int(n[child]), cmn, cmx, level + 1))
    assert seen == m.node_count + m.leaf_count and deepest == m.depth


@pytest.mark.parametrize("name,w,h", [("mesh_gorilla", 480, 270), ("mesh_cow", 480, 270), ("soup:200000", 384, 216)])
def test_gpu_built_tree_changes_few_hit_identities(ctx, name, w, h, record_property):
    cam = rt.default_camera(w / h)
    p = rt.make_params(w, h, spp=1, max_depth=5, light_samples=35, shadow_mode=1, jitter=0, seed=0)
    frames = []
    for mode, depth in ((0, 18 if name.startswith("soup") else 12), (2, 12)):
        ctx.upload_scene(rt.HostScene(name, 3, depth, tree_mode=mode))
        frames.append(ctx.render(cam, p))
    (rgb0, prim0, st0), (rgb1, prim1, st1) = frames
    changed = int(np.any(prim0 != prim1, axis=2).sum())
    dq = np.abs(np.minimum(np.nan_to_num(rgb0), 1.0) - np.minimum(np.nan_to_num(rgb1), 1.0)).max(axis=2)
    differ = int((dq > 1.0 / 255.0).sum())
    record_property("pixels_with_other_primary_hit", changed)
    print(f"{name}: {changed} of {w * h} pixels change their primary-hit identity under the GPU-built tree, {differ} their 8-bit radiance")
    assert st0.rays_primary == st1.rays_primary
    assert changed <= 0.002 * w * h and differ <= 0.01 * w * h


def test_gpu_build_of_a_large_soup_is_fast(record_property):
    t0 = time.time()
    scene = rt.HostScene("soup:2000000", 3, 24, tree_mode=2)
    t_gpu = time.time() - t0
    m = scene.arrays()["meshes"][0]
    record_property("seconds_scene_with_gpu_tree", t_gpu)
    print(f"2 M-triangle soup: scene set-up with the GPU-built tree {t_gpu:.2f} s, depth {m.depth}")
    assert m.depth <= 40 and m.leaf_count == 2000000
