"""Regenerates the data files under assets/ (repo root) from the reference checkout.

Run in the build container only (needs /root/reference):  python tests/golden/make_fixtures.py

models/*.off are the reference's own mesh DATA files (assets/models), copied
byte for byte: they are inputs of BASELINE configs 1, 2 and 4 (Scene.h:867,913,771).
img/normalMaps/n3.ppm and img/sphereTextures/s4.ppm are two of the reference's image DATA files
(assets/img): the normal map of setup_reflexion_chamber's floor (Scene.h:678) and a sphere texture of
setup_cornell_box (Scene.h:446), inputs of the texture / normal-map path (Material.h:86-138).
No reference source text is copied.
"""
import os
import shutil

REF = "/root/reference/assets/models"
HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "assets")
FILES = ["flamingo.off", "gorilla18_fixed.obj.off", "cow.off", "triangle.off", "cube.off"]
IMG_REF = "/root/reference/assets/img"
IMAGES = ["normalMaps/n3.ppm", "sphereTextures/s4.ppm"]

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

if __name__ == "__main__":
    main()
