"""rt_trace_static (the entry point that allocates nothing, include/rtgpu.h) against rt_trace, through the C-ABI:
two calls in a row on one context, the second with a smaller input, radiance bit for bit; limits are refused."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rays(rt, ctx, w, h, spp, seed):
    p = rt.make_params(w, h, spp=spp, max_depth=6, light_samples=35, shadow_mode=0, jitter=1, seed=seed)
    ctx.set_camera(rt.default_camera(w / h))
    org, dirs = ctx.camera_rays(p, 0, 0, w, h)
    n = len(org)
    pixel = (np.arange(n, dtype=np.uint32) // spp).astype(np.uint32)
    sample = (np.arange(n, dtype=np.uint32) % spp).astype(np.uint32)
    return p, org, dirs, pixel, sample


@pytest.mark.parametrize("scene", ["mesh_gorilla", "refraction", "reflexion_chamber"])
def test_static_trace_equals_rt_trace_twice(rt, ctx, scene):
    ctx.upload_scene(rt.HostScene(scene, 3, 12))
    # first call: 4096 rays (the limit); second call: fewer rays, other pixels, on the same scratch contents
    for (w, h, spp, seed) in [(64, 32, 2, 3), (24, 20, 3, 9)]:
        p, org, dirs, pixel, sample = _rays(rt, ctx, w, h, spp, seed)
        want = ctx.trace(org, dirs, p, pixel, sample)
        got = ctx.trace(org, dirs, p, pixel, sample, static=True)
        assert got.shape == want.shape
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{scene} {w}x{h}x{spp}: radiance bits differ"
        assert np.isfinite(want).all() and float(np.abs(want).max()) > 0.0


def test_static_trace_refuses_what_it_cannot_hold(rt, ctx):
    ctx.upload_scene(rt.HostScene("refraction", 3, 12))
    p, org, dirs, pixel, sample = _rays(rt, ctx, 80, 52, 1, 1)          # 4160 rays > 4096
    with pytest.raises(rt.RtError, match="limit"):
        ctx.trace(org, dirs, p, pixel, sample, static=True)
    p2 = rt.make_params(8, 8, spp=1, max_depth=9)                       # depth 9 > 8
    with pytest.raises(rt.RtError, match="limit"):
        ctx.trace(org[:16], dirs[:16], p2, pixel[:16], sample[:16], static=True)
    # and the context still works afterwards
    p3, org, dirs, pixel, sample = _rays(rt, ctx, 16, 16, 1, 1)
    a = ctx.trace(org, dirs, p3, pixel, sample)
    b = ctx.trace(org, dirs, p3, pixel, sample, static=True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
