"""The one output of the REAL reference this repository can be held against: its committed rendu.ppm.

rendu.ppm (reference repository root) is a 480x480 frame of scene 2, setup_cornell_box (Scene.h:288-466, main.cpp:390),
at the default camera (main.cpp:385) with nsamples = 30 (main.cpp:189), written by the P3 writer of main.cpp:245-255.
tests/golden/rendu_cornell_480.npz holds its pixels (tests/golden/make_fixtures.py). The reference draws its pixel jitter
and light samples from glibc rand() on racing threads, so single pixels carry soft-shadow noise that no re-run of the
reference itself would reproduce; what can be compared is everything that is not noise:

* geometry, pixel for pixel: camera unprojection, Square / Sphere intersection, u,v, the image and normal-map lookups,
  Phong under the coloured light, the 8-bit conversion -- on the surfaces whose inputs the reference checkout holds
  (back, left and right walls, floor, the textured sphere: 159 k of the 230 k pixels) the MEDIAN absolute difference
  is 0-2 levels and the mean signed difference per surface is a fraction of a level (no bias);
* the soft-shadow statistics: 8x8 block means on those surfaces agree to a fraction of a level.

Not comparable: the ceiling -- the author's checkout had normalMaps/sand.png, this one does not (a failed load leaves
the material plain, Material.h:134-138) -- and, through it, what the mirror and glass spheres reflect of the ceiling.

* the glass and the mirror sphere WHERE THEY SHOW NOTHING OF THE CEILING (round 3). Which pixels those are is decided by
  rendering the room a second time with a ceiling that returns no light (oracle scene "cornell_box_black_ceiling"): a pixel
  with the same bits in both frames carries no contribution of the ceiling's shading. That leaves 17.8 k of the glass
  sphere's 20.1 k pixels and 12.6 k of the mirror's 14.2 k, and on them the frames agree with rendu.ppm like on the walls:
  glass median |delta| 2 levels, mean signed difference <= 0.07 of a level, 83 % within 4 levels; mirror median 1,
  <= 0.09, 92 %. That is Material::reflect / refract / reflectance (Material.h:60-84), the absorption and weighting of
  Scene.h:225-272 and the recursion held against the real program's pixels. Two controls show what the comparison can
  see: a glass of index 1.3 instead of 1.2 (median 15 levels, 18 % within 4) and a mirror of reflectivity 0.6 instead of
  0.5 (mean signed difference 5.8 levels) fail it.

This is a statistical pin of the oracle (and of the HIP path) to the reference's own output, not bit parity: DESIGN 6.
"""
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
COMPARABLE = [0, 1, 2, 3, 8]      # back wall, left wall, right wall, floor, textured sphere (object ordinals: squares, then spheres)


def reference_frame():
    return np.load(os.path.join(HERE, "golden", "rendu_cornell_480.npz"))["rgb"].astype(np.int64)


def hold_against_reference(rgb, prim_obj):
    ref = reference_frame()
    q = O.quantize(rgb).astype(np.int64)
    assert q.shape == ref.shape == (480, 480, 3)
    d = np.abs(q - ref).max(axis=2)
    seen = set(np.unique(prim_obj).tolist())
    assert seen == {0, 1, 2, 3, 4, 6, 7, 8}, seen            # front wall (5) faces away; every other object is in view
    report = {}
    for o in COMPARABLE:
        m = prim_obj == o
        bias = (q - ref)[m].mean(axis=0)
        report[o] = (int(m.sum()), float(np.median(d[m])), float(d[m].mean()), bias.round(3).tolist())
        assert np.median(d[m]) <= 2, report
        assert np.abs(bias).max() < 0.75, report              # measured: <= 0.29 of a level
    keep = np.isin(prim_obj, COMPARABLE)
    assert keep.sum() > 150000
    # geometry: most comparable pixels are within soft-shadow noise of the reference's
    assert (d[keep] <= 4).mean() > 0.85, report
    blocks = lambda a: a.reshape(60, 8, 60, 8, -1).mean(axis=(1, 3))
    whole = blocks(keep.astype(np.float64)[..., None])[..., 0] == 1.0
    bd = np.abs(blocks(q.astype(np.float64)) - blocks(ref.astype(np.float64))).max(axis=2)[whole]
    assert whole.sum() > 2000 and bd.mean() < 0.75 and np.percentile(bd, 99) < 5.0, (bd.mean(), np.percentile(bd, 99))
    return report


_frames = {}


def oracle_frame(name="cornell_box", window=None):
    """The oracle's frame of a variant of the room (cached: the CPU suite renders each once), whole or a window (x0, y0, w, h)."""
    key = (name, window)
    if key not in _frames:
        prm = O.make_params(480, 480, spp=30, jitter=1, seed=1)     # nsamples = 30, main.cpp:189; depth 50, Scene.h:279
        args = () if window is None else tuple(int(v) for v in window)
        rgb, prim, st, _ = O.Scene(name, 3, 12).render(O.default_camera(1.0), prm, *args, threads=min(8, os.cpu_count() or 1))
        _frames[key] = (rgb, prim, st)
    return _frames[key]


def free_of_the_ceiling():
    """Pixels whose bits do not change when the ceiling returns no light."""
    a, b = oracle_frame()[0], oracle_frame("cornell_box_black_ceiling")[0]
    return (np.ascontiguousarray(a).view(np.uint32) == np.ascontiguousarray(b).view(np.uint32)).all(axis=2)


GLASS, MIRROR = 6, 7            # object ordinals (Scene.h:408-435)


def sphere_figures(rgb, prim_obj, which, mask):
    ref = reference_frame()
    q = O.quantize(rgb).astype(np.int64)
    m = (prim_obj == which) & mask
    d = np.abs(q - ref).max(axis=2)[m]
    return int(m.sum()), float(np.median(d)), float(d.mean()), (q - ref)[m].mean(axis=0), float((d <= 4).mean())


def hold_spheres_against_reference(rgb, prim_obj):
    mask = free_of_the_ceiling()
    n, med, mean, bias, within = sphere_figures(rgb, prim_obj, GLASS, mask)
    assert n > 15000 and med <= 3 and np.abs(bias).max() < 0.75 and within > 0.75, ("glass", n, med, mean, bias, within)
    n, med, mean, bias, within = sphere_figures(rgb, prim_obj, MIRROR, mask)
    assert n > 11000 and med <= 2 and np.abs(bias).max() < 0.75 and within > 0.85, ("mirror", n, med, mean, bias, within)


def test_oracle_frame_of_the_cornell_box_against_the_references_rendu_ppm():
    rgb, prim, st = oracle_frame()
    hold_against_reference(rgb, prim[..., 0])


def test_glass_and_mirror_spheres_of_the_oracle_frame_against_rendu_ppm():
    rgb, prim, st = oracle_frame()
    hold_spheres_against_reference(rgb, prim[..., 0])
    # controls: the same comparison with other optics must fail (rendered on the sphere's bounding window only)
    obj = prim[..., 0]
    mask = free_of_the_ceiling()
    for name, which in (("cornell_box_ior_1p3", GLASS), ("cornell_box_reflectivity_0p6", MIRROR)):
        ys, xs = np.nonzero(obj == which)
        x0, x1, y0, y1 = xs.min(), xs.max() + 1, ys.min(), ys.max() + 1
        part = oracle_frame(name, (x0, y0, x1 - x0, y1 - y0))[0]
        other = rgb.copy()
        other[y0:y1, x0:x1] = part
        n, med, mean, bias, within = sphere_figures(other, obj, which, mask)
        if which == GLASS:
            assert med >= 8 and within < 0.4, (name, med, mean, bias, within)
        else:
            assert np.abs(bias).max() > 3.0 and within < 0.65, (name, med, mean, bias, within)


@pytest.mark.gpu
def test_gpu_frame_of_the_cornell_box_against_the_references_rendu_ppm(rt, ctx):
    hs = rt.HostScene("cornell_box", 3, 12)
    ctx.upload_scene(hs)
    rgb, prim, st = ctx.render(rt.default_camera(1.0), rt.make_params(480, 480, spp=30, jitter=1, seed=1))
    obj = prim[..., 0].astype(np.int64)
    obj[obj == 0xFFFFFFFF] = -1
    hold_against_reference(rgb, obj)
    hold_spheres_against_reference(np.nan_to_num(rgb, nan=0.0, posinf=1.0), obj)      # glass and mirror, where nothing of the ceiling shows
    # and against the oracle's frame of the same seed: the usual bar (1/255 on the 8-bit value). Unnormalised normals from
    # the normal maps (Scene.h:199-202) drive single specular samples beyond 6e7, which the fixed-point accumulators
    # report as infinite (deviation 3): such pixels are saturated in the oracle's frame as well.
    orgb, oprim, ost = oracle_frame()
    assert np.array_equal(obj, oprim[..., 0])
    bad = ~np.isfinite(rgb).all(axis=2)
    assert 0 < (~np.isfinite(rgb)).sum() <= min(st.nonfinite, 256)   # the counter counts contributions, the frame pixels
    assert (orgb[bad].max(axis=1) >= 1e5).all()
    assert np.abs(O.quantize(np.nan_to_num(rgb, nan=0.0, posinf=1.0)) - O.quantize(orgb)).max() <= 1
    assert (st.rays_primary, st.rays_secondary) == (ost.rays_primary, ost.rays_secondary)
