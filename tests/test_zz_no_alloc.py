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


def test_static_trace_leaves_free_device_memory_unchanged(rt, ctx):
    """The property the entry point exists for: a (warm) call allocates and frees nothing on the device. hipMemGetInfo is read
    through the HIP runtime the library is bound to (by soname, as tests/test_progressive.py does); rt_trace, which stages
    its inputs in buffers of its own, serves as the check that the probe sees an allocation when there is one."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so.7")
    hip.hipMemGetInfo.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]

    def free_bytes():
        f, t = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value

    ctx.upload_scene(rt.HostScene("mesh_gorilla", 3, 12))
    p, org, dirs, pixel, sample = _rays(rt, ctx, 64, 32, 2, 5)
    want = ctx.trace(org, dirs, p, pixel, sample)
    ctx.trace(org, dirs, p, pixel, sample, static=True)             # warm: code objects loaded, scratch address taken
    before = free_bytes()
    got = ctx.trace(org, dirs, p, pixel, sample, static=True)
    p2, org2, dirs2, pixel2, sample2 = _rays(rt, ctx, 24, 20, 3, 9)
    ctx.trace(org2, dirs2, p2, pixel2, sample2, static=True)
    assert free_bytes() == before, "rt_trace_static changed the free device memory"
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


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
