"""Progressive accumulation (rt_continue_frame, the GL-free part of SURVEY 8(f).4): passes that add samples to one frame
leave the bits of the frame rendered in one go -- same sample numbers, same counter-RNG streams, order-independent sums --
and every intermediate resolve is the frame of the samples taken so far."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene,depth", [("cornell_box", 50), ("mesh_cow", 8), ("final", 6)])
def test_passes_add_up_to_the_one_shot_frame(rt, ctx, scene, depth):
    hs = rt.HostScene(scene, 3, 12)
    ctx.upload_scene(hs)
    W, H = 96, 64
    cam = rt.default_camera(W / H)
    full, fprim, fst = ctx.render(cam, rt.make_params(W, H, spp=16, max_depth=depth, jitter=1, seed=9))
    first4, _, _ = ctx.render(cam, rt.make_params(W, H, spp=4, max_depth=depth, jitter=1, seed=9))
    tiles = rt.interleaved_tiles(W, H)
    ctx.set_camera(cam)
    ctx.begin_frame(rt.make_params(W, H, spp=4, max_depth=depth, jitter=1, seed=9))
    for t in tiles:
        ctx.render_tile(*t)
    rgb, prim, _ = ctx.download()
    assert np.array_equal(rgb.view(np.uint32), first4.view(np.uint32)), "the first pass is the 4-sample frame"
    for more in (4, 8):
        ctx.continue_frame(more)
        for t in tiles[::-1]:                      # any tile order
            ctx.render_tile(*t)
    rgb, prim, st = ctx.download()
    assert np.array_equal(rgb.view(np.uint32), full.view(np.uint32)), "4 + 4 + 8 samples differ from 16 at once"
    assert np.array_equal(prim, fprim)
    assert (st.rays_primary, st.rays_secondary, st.rays_shadow) == (fst.rays_primary, fst.rays_secondary, fst.rays_shadow)


def test_continue_frame_needs_an_open_frame_and_a_sane_count(rt):
    c = rt.Context(0)
    try:
        c.upload_scene(rt.HostScene("refraction", 3, 12))
        with pytest.raises(rt.RtError):
            c.continue_frame(4)
        c.set_camera(rt.default_camera(1.0))
        c.begin_frame(rt.make_params(32, 32, spp=2))
        with pytest.raises(rt.RtError):
            c.continue_frame(0)
        c.continue_frame(3)
    finally:
        c.close()


@pytest.mark.parametrize("scene", ["cornell_box", "reflexion_chamber"])
def test_displayable_frame_is_the_ppm_bytes(rt, ctx, scene):
    """rt_resolve_rgba8_device (the device half of SURVEY 8(f).4): RGBA8 in caller-owned device memory = the bytes of the PPM
    (main.cpp:251-253), top-down or bottom-up, after every progressive pass."""
    import ctypes
    # caller-owned device memory straight from the HIP runtime the library itself uses: by soname, so that the copy
    # librtgpu.so is already bound to comes back (/opt/rocm's, or torch's bundled one if torch was imported first)
    hip = ctypes.CDLL("libamdhip64.so.7")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    hs = rt.HostScene(scene, 3, 12)
    ctx.upload_scene(hs)
    W, H = 80, 48
    ctx.set_camera(rt.default_camera(W / H))
    ctx.begin_frame(rt.make_params(W, H, spp=2, max_depth=8, jitter=1, seed=4))
    dptr = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dptr), W * H * 4) == 0
    host = np.zeros((H, W, 4), np.uint8)
    for more in (0, 3):
        if more:
            ctx.continue_frame(more)
        for t in rt.interleaved_tiles(W, H):
            ctx.render_tile(*t)
        rgb, _, _ = ctx.download()
        m = np.where(rgb < 1.0, rgb, np.float32(1.0)).astype(np.float32)            # std::min<float>(1.f, c) keeps 1 for a NaN
        v = (np.float32(255.0) * m).astype(np.float32)
        want = np.where(v >= 0, v, 0).astype(np.int32).reshape(H, W, 3)
        assert want.max() <= 255
        for bottom_up in (False, True):
            assert hip.hipMemset(dptr, 0, W * H * 4) == 0
            ctx.resolve_rgba8_device(dptr.value, bottom_up)
            assert hip.hipMemcpy(host.ctypes.data, dptr, W * H * 4, 2) == 0          # hipMemcpyDeviceToHost
            got = host.copy()
            if bottom_up:
                got = got[::-1]
            assert np.array_equal(got[..., :3].astype(np.int32), want)
            assert (got[..., 3] == 255).all()
    assert want.min() < 255 and want.max() > 0
    hip.hipFree(dptr)
