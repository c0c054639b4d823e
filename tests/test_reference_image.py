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


def test_oracle_frame_of_the_cornell_box_against_the_references_rendu_ppm():
    sc = O.Scene("cornell_box", 3, 12)
    prm = O.make_params(480, 480, spp=30, jitter=1, seed=1)     # nsamples = 30, main.cpp:189; depth 50, Scene.h:279
    rgb, prim, st, _ = sc.render(O.default_camera(1.0), prm, threads=min(8, os.cpu_count() or 1))
    hold_against_reference(rgb, prim[..., 0])


@pytest.mark.gpu
def test_gpu_frame_of_the_cornell_box_against_the_references_rendu_ppm(rt, ctx):
    hs = rt.HostScene("cornell_box", 3, 12)
    ctx.upload_scene(hs)
    rgb, prim, st = ctx.render(rt.default_camera(1.0), rt.make_params(480, 480, spp=30, jitter=1, seed=1))
    obj = prim[..., 0].astype(np.int64)
    obj[obj == 0xFFFFFFFF] = -1
    hold_against_reference(rgb, obj)
    # and against the oracle's frame of the same seed: the usual bar (1/255 on the 8-bit value). Unnormalised normals from
    # the normal maps (Scene.h:199-202) drive single specular samples beyond 6e7, which the fixed-point accumulators
    # report as infinite (deviation 3): such pixels are saturated in the oracle's frame as well.
    orgb, oprim, ost, _ = O.Scene("cornell_box", 3, 12).render(O.default_camera(1.0), O.make_params(480, 480, spp=30, jitter=1, seed=1), threads=8)
    assert np.array_equal(obj, oprim[..., 0])
    bad = ~np.isfinite(rgb).all(axis=2)
    assert 0 < (~np.isfinite(rgb)).sum() <= min(st.nonfinite, 256)   # the counter counts contributions, the frame pixels
    assert (orgb[bad].max(axis=1) >= 1e5).all()
    assert np.abs(O.quantize(np.nan_to_num(rgb, nan=0.0, posinf=1.0)) - O.quantize(orgb)).max() <= 1
    assert (st.rays_primary, st.rays_secondary) == (ost.rays_primary, ost.rays_secondary)
