"""BASELINE config 5 at its full size -- the 10 M-triangle soup, mesh tree depth 24, a 3840x2160 frame at 16 spp with
soft shadows -- through properties that do not depend on the size: a centred crop against the oracle, the conservative
shadow tree against the exact walk on the same crop, and disjoint tile sets against one render."""
import numpy as np
import pytest

import oracle_lib as O
from test_gpu_parity import bits, radiance_check

pytestmark = pytest.mark.gpu

W, H, SPP, DEPTH = 3840, 2160, 16, 5
NAME, MESH_DEPTH = "soup:10000000", 24


@pytest.mark.timeout(600)
def test_config5_full_size_properties(rt, ctx, monkeypatch):
    hs = rt.HostScene(NAME, 3, MESH_DEPTH)
    assert hs.desc.contents.n_tris == 10_000_000
    ctx.upload_scene(hs)
    assert ctx.L.rt_uses_quantised_shadow_nodes(ctx.h) == 1
    cam = rt.default_camera(W / H)
    p = rt.make_params(W, H, spp=SPP, max_depth=DEPTH, light_samples=35, shadow_mode=0, jitter=1, seed=0)
    c = 64
    x0, y0 = (W - c) // 2 // 32 * 32, (H - c) // 2 // 32 * 32          # tile-aligned centred crop
    tiles = [(x0 + dx, y0 + dy, 32, 32) for dy in (0, 32) for dx in (0, 32)]
    rgb, prim, st = ctx.render(cam, p, tiles)
    crop, pcrop = rgb[y0:y0 + c, x0:x0 + c], prim[y0:y0 + c, x0:x0 + c]
    assert st.rays_primary == c * c * SPP and st.rays_shadow > 10 * st.rays_primary and st.nonfinite == 0
    assert not rgb[:y0].any() and not rgb[y0 + c:].any()                # nothing outside the queued tiles

    # 1. disjoint tile sets (two "ranks") give the same pixels
    union = np.zeros_like(crop)
    shadow_rays = 0
    for part in (tiles[0::2], tiles[1::2]):
        r2, _, s2 = ctx.render(cam, p, part)
        for (tx, ty, tw, th) in part:
            union[ty - y0:ty - y0 + th, tx - x0:tx - x0 + tw] = r2[ty:ty + th, tx:tx + tw]
        shadow_rays += s2.rays_shadow
    assert np.array_equal(bits(union), bits(crop)) and shadow_rays == st.rays_shadow

    # 2. the exact 64-byte nodes instead of the quantised 4-wide tree: same bits
    monkeypatch.setenv("RT_SHADOW_EXACT_NODES", "1")
    c2 = rt.Context(0)
    try:
        c2.upload_scene(hs)
        assert c2.L.rt_uses_quantised_shadow_nodes(c2.h) == 0
        ergb, eprim, est = c2.render(cam, p, tiles)
    finally:
        c2.close()
        monkeypatch.delenv("RT_SHADOW_EXACT_NODES")
    assert np.array_equal(bits(ergb[y0:y0 + c, x0:x0 + c]), bits(crop)) and np.array_equal(eprim, prim)
    assert est.rays_shadow == st.rays_shadow

    # 3. the oracle on the crop: primary hits exact, ray counts equal, radiance within 1/255 of the 8-bit value
    osc = O.Scene(NAME, 3, MESH_DEPTH)
    oc = O.default_camera(W / H)
    op = O.make_params(W, H, spp=SPP, max_depth=DEPTH, light_samples=35, shadow_mode=0, jitter=1, seed=0)
    orgb, oprim, ost, _ = osc.render(oc, op, x0, y0, c, c, threads=16)
    oprim_u = oprim.astype(np.int64)
    oprim_u[oprim[..., 0] < 0] = 0xFFFFFFFF
    assert np.array_equal(pcrop.astype(np.int64), oprim_u)
    assert (st.rays_primary, st.rays_secondary, st.rays_shadow) == (ost.rays_primary, ost.rays_secondary, ost.rays_shadow)
    radiance_check(crop, orgb, "config 5 crop")
