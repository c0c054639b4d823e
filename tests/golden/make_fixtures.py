"""Regenerates the data files under assets/ (repo root) from the reference checkout.

Run in the build container only (needs /root/reference):  python tests/golden/make_fixtures.py

models/*.off are the reference's own mesh DATA files (assets/models), copied
byte for byte: they are inputs of BASELINE configs 1, 2 and 4 (Scene.h:867,913,771).
img/** are the reference's image DATA files (assets/img) that its scenes load: the normal map of
setup_reflexion_chamber's floor (Scene.h:678) and the images setup_cornell_box finds in the reference
checkout (Scene.h:326-446: n4.ppm, normalMaps/cliff.png, s5.ppm, normalMaps/stone_floor.png, sand.jpg,
s3.ppm, s4.ppm; cliff.jpg, stone_floor.jpg and normalMaps/sand.png are not in the checkout, so those
loads fail there and the material stays untextured). PPM and PNG files are copied byte for byte.
textures/sand.ppm holds the texels the REFERENCE'S OWN decoder (its vendored stb_image.h, compiled
unmodified into oracle/_ref/libref.so) returns for textures/sand.jpg: a JPEG decoder's output depends
on its IDCT and chroma filter, so the decoded texels are the fixture, not the file.
golden/rendu_cornell_480.npz is the reference's committed output image rendu.ppm (the cornell box,
480x480, 30 spp, main.cpp:189, default camera) as a uint8 array: the one output of the real
reference this repository can compare against (tests/test_reference_image.py).
golden/stage_*.npz are three more committed output frames of the reference, the ones that reach a triangle mesh:
etapes/normal_interpolation/with.ppm and without.ppm (blob.off in the room of setup_test_scene, default camera) and
etapes/mesh/gorilla_1.ppm (the gorilla in the same room, camera turned by the mouse). The reference no longer holds the
scene blocks that made them, so what Scene.h does not say was recovered from the pixels (scripts kept under scratch/ while
fitting, the recipe in short):
* light: least squares of the Phong formula of Scene.h:90-108 over the back wall's 60 k pixels -> (0.001, 3.472, -0.039),
  rms 1.3 levels: the light of Scene.h:477, (0, 3.5, 0); hard shadows (radius 0.5 leaves 90 % of the floor within 2 levels,
  radius 0 98.5 %);
* blob: silhouette IoU over the 24 axis-aligned orientations, then the mean absolute difference to without.ppm over
  rotation vector, scale and translation (Nelder-Mead on oracle renders) -> rotation 0.0007 rad off rotate_x(180), scale
  1.0001, translation (-0.1251, -1.0501, -1.253): rotate_x(180), translate(-0.125, -1.05, -1.25);
* with.ppm was written when Mesh.cpp:117-130 did not yet normalise the interpolated normal (the same weights and vertex
  normals without the final normalize() put 95.6 % of the mesh pixels within 2 levels, with it 75 %);
* gorilla_1.ppm: camera fitted on the wall / floor / background masks (trackball rotation + translation).
No reference source text is copied.
"""
import ctypes
import os
import shutil

import numpy as np

REF = "/root/reference/assets/models"
HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "assets")
FILES = ["flamingo.off", "gorilla18_fixed.obj.off", "cow.off", "triangle.off", "cube.off", "blob.off"]
# committed OUTPUT frames of the real reference that reach the mesh path (P3 text -> uint8 arrays): tests/test_stage_images.py
STAGE_IMAGES = [("etapes/normal_interpolation/with.ppm", "stage_blob_with_480.npz"),
                ("etapes/normal_interpolation/without.ppm", "stage_blob_without_480.npz"),
                ("etapes/mesh/gorilla_1.ppm", "stage_gorilla_1_480.npz")]
IMG_REF = "/root/reference/assets/img"
IMAGES = ["normalMaps/n3.ppm", "sphereTextures/s4.ppm", "normalMaps/n4.ppm", "normalMaps/cliff.png",
          "normalMaps/stone_floor.png", "sphereTextures/s5.ppm", "sphereTextures/s3.ppm"]
DECODED = [("textures/sand.jpg", "textures/sand.ppm")]
REPO = os.path.dirname(HERE)


def stbi_load(path):
    """The reference's decoder (oracle/_ref/libref.so, built from /root/reference/include/stb_image.h by oracle/Makefile)."""
    L = ctypes.CDLL(os.path.join(REPO, "oracle", "_ref", "libref.so"))
    L.ref_stbi_load.argtypes = [ctypes.c_char_p] + [ctypes.POINTER(ctypes.c_int)] * 3 + [ctypes.c_void_p, ctypes.c_long]
    w, h, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert L.ref_stbi_load(path.encode(), w, h, c, None, 0) == 0, path
    px = np.empty((h.value, w.value, c.value), dtype=np.uint8)
    assert L.ref_stbi_load(path.encode(), w, h, c, px.ctypes.data, px.size) == 0
    return px


def main():
    os.makedirs(os.path.join(HERE, "models"), exist_ok=True)
    for f in FILES:
        shutil.copyfile(os.path.join(REF, f), os.path.join(HERE, "models", f))
        print("copied", f)
    for f in IMAGES:
        dst = os.path.join(HERE, "img", f)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(IMG_REF, f), dst)
        print("copied", f)
    for src, dst in DECODED:
        px = stbi_load(os.path.join(IMG_REF, src))
        assert px.shape[2] == 3
        out = os.path.join(HERE, "img", dst)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        with open(out, "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (px.shape[1], px.shape[0]))
            f.write(px.tobytes())
        print("decoded", src, "->", dst, px.shape)
    tok = open("/root/reference/rendu.ppm").read().split()
    assert tok[0] == "P3" and tok[3] == "255"
    w, h = int(tok[1]), int(tok[2])
    img = np.array(tok[4:4 + 3 * w * h], dtype=np.uint8).reshape(h, w, 3)
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "rendu_cornell_480.npz"), rgb=img)
    print("rendu.ppm ->", img.shape)
    for src, dst in STAGE_IMAGES:
        tok = open(os.path.join("/root/reference", src)).read().split()
        assert tok[0] == "P3" and tok[3] == "255"
        w, h = int(tok[1]), int(tok[2])
        img = np.array(tok[4:4 + 3 * w * h], dtype=np.uint8).reshape(h, w, 3)
        np.savez_compressed(os.path.join(REPO, "tests", "golden", dst), rgb=img)
        print(src, "->", dst, img.shape)

if __name__ == "__main__":
    main()
