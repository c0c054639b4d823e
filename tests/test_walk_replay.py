"""The GPU's walks replayed on the CPU under AddressSanitizer (tools/emu/shadow_emu.cpp: rt_device.h / rt_kernels.h compiled with g++ behind a
shim of the HIP names) on a device scene the library dumped on an MI355X (RT_DUMP_SCENE; tests/golden/emu_scene_102 = random scene 102 of
tests/test_gpu_parity.py, a mirror mesh with a degenerate triangle under fat leaves): no walk of k_shadow, k_closest or k_intersect may read or
write outside its buffers or its stack -- rays with NaNs, infinities and zero directions included, which is what faulted the GPU until round 3 --
and the conservative walks must give the exact walk's answers. No GPU needed; the fixture is the library's own output, not the reference's."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "emu_scene_102")


@pytest.fixture(scope="module")
def emulator(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("emu") / "shadow_emu")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address", "-static-libasan", "-ffp-contract=off", "-DRT_HOST_EMU", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "ray-tracing_amd", "csrc"), "-I" + os.path.join(ROOT, "tools", "emu"), "-Wno-attributes", "-o", exe,
           os.path.join(ROOT, "tools", "emu", "shadow_emu.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


def test_walks_stay_inside_their_buffers_and_agree_with_the_exact_walk(emulator):
    mode = 0            # the kernel variant the library picks for the scene (k_shadow<2>: fat leaves with sub-trees)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0")      # (statically linked runtime; a preloaded library in the environment must not stop it)
    r = subprocess.run([emulator, FIXTURE, str(mode), "2", "2"], capture_output=True, text=True, timeout=900, env=env)
    if r.returncode == 3:
        pytest.skip("the fixture predates a change of the DScene layout: " + r.stderr.strip()[-200:])
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert " 0 differ between the binary and the 4-wide walk" in r.stdout
    assert "overflow flag 0, 0 answers differ from the exact walk" in r.stdout
