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
