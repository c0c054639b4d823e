"""The reference-side binding shipped as code (integration/rt_from_reference.h, integration/main_gpu.patch): the header's
templates, instantiated with stand-in types that carry the member names of the reference's classes (private members and the
one `friend` line of the patch included, integration/test/ref_like.h), flatten every named scene to the bytes the host mirror's
rth::Scene::flatten() produces -- trees rebuilt as BVH_Node pointer trees, BuildingMesh::ref pointing into the scene's own
vectors, texel pointers in the materials. Host only: no GPU call is made."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rt_flatten_over_reference_named_types_gives_the_host_mirrors_bytes():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "integration")], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "integration", "build", "test_flatten")
    out = subprocess.run([exe, os.path.join(ROOT, "assets", "models"), os.path.join(ROOT, "assets", "img")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 10 and all("same bytes" in l for l in lines), out.stdout


def test_the_patch_only_adds():
    """integration/main_gpu.patch leaves the reference's render loop in place (the GPU path is tried first and falls back to it):
    apart from the two Makefile lines and the empty line of clear() it removes nothing."""
    removed = [l for l in open(os.path.join(ROOT, "integration", "main_gpu.patch"), newline="").read().split("\n")
               if l.startswith("-") and not l.startswith("---")]
    assert len(removed) == 3 and sum("LIBS" in l or "CPPFLAGS" in l for l in removed) == 2, removed
    text = open(os.path.join(ROOT, "integration", "main_gpu.patch")).read()
    for needle in ("friend struct rtref::Access;", "rtref::rt_flatten(scenes[selected_scene], flat)", "rtref::rt_pack_camera(camera)", "rt_download_framebuffer",
                   "if (g_rt && ray_trace_from_camera_gpu()) return;"):
        assert needle in text, needle
