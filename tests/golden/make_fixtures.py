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
No reference source text is copied.
"""
import ctypes
import os
import shutil

import numpy as np

REF = "/root/reference/assets/models"
HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "assets")
FILES = ["flamingo.off", "gorilla18_fixed.obj.off", "cow.off", "triangle.off", "cube.off"]
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

if __name__ == "__main__":
    main()
