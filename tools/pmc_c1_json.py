#!/usr/bin/env python3
"""usage: tools/pmc_c1_json.py gpurun_out/<tag> <round-tag>
Turns the raw sums of tools/pmc_c1.sh into profiles/c1_k_closest_pmc.json (per-ray figures of k_closest on C1, which bench.py
prices side_configs.C1's bounce kernel with) and profiles/<round-tag>_pmc_c1.txt."""
import hashlib
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = {}
for name in ("sq", "fetch", "write", "tc"):
    raw.update(json.load(open(os.path.join(src, f"pmc_{name}.json"))))
plain = json.loads(open(os.path.join(src, "plain.json")).read().strip().splitlines()[-1])
rays = plain["rays_per_frame"]["primary"] + plain["rays_per_frame"]["secondary"]       # what k_closest traces in one frame
launches = raw["SQ_WAVE_CYCLES"]["launches"]
h = hashlib.sha256()
for f in ("ray-tracing_amd/csrc/rt_kernels.h", "ray-tracing_amd/csrc/rt_device.h", "ray-tracing_amd/csrc/rtgpu.hip"):
    h.update(open(os.path.join(ROOT, f), "rb").read())
tcv, act, ins, wc = (raw[k]["sum"] for k in ("SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES"))
hbm = (2 * raw["FETCH_SIZE"]["sum"] + raw["WRITE_SIZE"]["sum"]) * 1024
rec = {"config": "C1: default scene 480x480, 30 spp, depth 50", "rays_per_frame": rays, "launches_per_frame": launches,
       "valu_lane_ops_per_ray": tcv / rays, "valu_insts_per_ray": ins / rays, "lanes_active_per_valu_inst": tcv / (64.0 * ins),
       "valu_busy_of_wave_cycles": act / wc, "hbm_bytes_per_ray": hbm / rays,
       "l1_lane_accesses_per_ray": raw["TCP_TOTAL_CACHE_ACCESSES_sum"]["sum"] / rays,
       "l2_hit": raw["TCC_HIT_sum"]["sum"] / (raw["TCC_HIT_sum"]["sum"] + raw["TCC_MISS_sum"]["sum"]),
       "k_closest_ms_per_frame_plain_run": plain["kernel_ms_per_frame"]["k_closest"], "ms_per_frame_plain_run": plain["ms_per_frame"],
       "kernel_source_sha256": h.hexdigest(),
       "source": f"profiles/{tag}_pmc_c1.txt (tools/pmc_c1.sh: rocprofv3 --pmc passes over one C1 frame, k_closest rows of all its launches)"}
json.dump(rec, open(os.path.join(ROOT, "profiles", "c1_k_closest_pmc.json"), "w"), indent=1)
with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_c1.txt"), "w") as f:
    f.write(f"rocprofv3 PMC passes over one C1 frame (default scene 480x480, 30 spp, depth 50), k_closest only, {launches} launches, {rays} primary + secondary rays\n")
    for k, v in sorted(raw.items()):
        f.write(f"{k:32s} sum {v['sum']:.6g}\n")
    f.write(f"=> VALU lane-ops / ray {tcv / rays:.1f} in {ins / rays:.1f} wave instructions; lanes active per VALU instruction {tcv / (64 * ins):.3f}; VALU busy share of wave cycles {act / wc:.3f}\n")
    f.write(f"=> HBM-side bytes / ray {hbm / rays:.1f}; L1 lane-accesses / ray {rec['l1_lane_accesses_per_ray']:.1f}; L2 hit {rec['l2_hit']:.3f}\n")
    f.write(f"plain run of the same build: {plain['ms_per_frame']} ms per frame, k_closest {plain['kernel_ms_per_frame']['k_closest']} ms, k_shadow {plain['kernel_ms_per_frame']['k_shadow']} ms\n")
print(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_c1.txt")).read())
