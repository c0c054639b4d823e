"""The roofline fractions of bench.py multiply live kernel times by per-ray counters from committed PMC runs (profiles/*.json). They describe
the build only while the kernel sources are the ones the counters were taken on: bench.py checks that at run time and quotes no fraction
otherwise; this test says so before a stale file is committed (rerun tools/pmc_c5.sh + tools/pmc_c5_json.py and tools/pmc_c1.sh +
tools/pmc_c1_json.py on an MI355X after changing rt_kernels.h, rt_device.h or rtgpu.hip)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_pmc_files_were_taken_on_these_kernel_sources():
    b = load_bench()
    h = b.kernel_source_hash()
    assert h is not None
    c5 = json.load(open(b.PMC_FILE))
    for key, rec in c5.items():
        assert rec.get("kernel_source_sha256") == h, f"profiles/c5_k_shadow_pmc.json[{key}] is stale: rerun tools/pmc_c5.sh"
    c1 = json.load(open(b.PMC_C1_FILE))
    assert c1.get("kernel_source_sha256") == h, "profiles/c1_k_closest_pmc.json is stale: rerun tools/pmc_c1.sh"


def test_roofline_entries_from_the_committed_counters():
    """The arithmetic of the two roofline objects on the committed counters: fractions in (0, 1), the flop figure by SURVEY 8(d)'s formula."""
    b = load_bench()
    r = b.closest_roofline(16_144_582, 11.7)
    assert r["pmc_matches_build"] and 0.0 < r["frac"] < 1.0 and r["kernel"] == "k_closest"
    pmc = json.load(open(b.PMC_FILE))["soup:10000000/d24/3840x2160x16"]
    ops = pmc["SQ_THREAD_CYCLES_VALU_per_launch"] / pmc["rays_per_launch"]
    assert 10_000 < ops < 20_000
    flops = pmc["node_steps_per_ray"] * 4 * 30 + pmc["tri_steps_per_ray"] * 160
    assert abs(flops - 14_777) < 50
    assert b.VALU_PEAK_TFLOPS == 2 * b.VALU_PEAK_TLANEOPS and abs(b.VALU_PEAK_TFLOPS - 157.3) < 0.1
