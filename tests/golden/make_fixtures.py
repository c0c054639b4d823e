"""Regenerates the data fixtures under tests/golden/ from the reference checkout.

Run in the build container only (needs /root/reference):  python tests/golden/make_fixtures.py

models/*.off are the reference's own mesh DATA files (assets/models), copied
byte for byte: they are inputs of BASELINE configs 1, 2 and 4 (Scene.h:867,913,771).
No reference source text is copied.
"""
import os
import shutil

REF = "/root/reference/assets/models"
HERE = os.path.dirname(os.path.abspath(__file__))
FILES = ["flamingo.off", "gorilla18_fixed.obj.off", "cow.off", "triangle.off", "cube.off"]

def main():
    os.makedirs(os.path.join(HERE, "models"), exist_ok=True)
    for f in FILES:
        shutil.copyfile(os.path.join(REF, f), os.path.join(HERE, "models", f))
        print("copied", f)

if __name__ == "__main__":
    main()
