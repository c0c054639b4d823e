#!/usr/bin/env python3
"""usage: tools/pmc_c5_json.py gpurun_out/<tag> <round-tag> [workload key]
Turns the raw sums of tools/pmc_c5.sh into profiles/c5_k_shadow_pmc.json (per-launch figures bench.py reads) and
profiles/<round-tag>_pmc_c5.txt (what a reader recomputes the bench line's roofline from)."""
import json
import os
import re
import sys

src, tag = sys.argv[1], sys.argv[2]
key = sys.argv[3] if len(sys.argv) > 3 else "soup:10000000/d24/3840x2160x16"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = {}
for name in ("sq", "fetch", "write", "tc"):
    raw.update(json.load(open(os.path.join(src, f"pmc_{name}.json"))))
bench = json.loads(open(os.path.join(src, "count.json")).read().strip().splitlines()[-1])
err = open(os.path.join(src, "count.err")).read()
m = re.search(r"shadow steps: node (\d+) tri (\d+) other (\d+) lane-iterations (\d+) rays (\d+); wave iterations: node (\d+) tri (\d+)", err)
node, tri, other, lane_it, rays, wn, wt = (int(x) for x in m.groups())
launches = raw["SQ_WAVE_CYCLES"]["launches"]
per = {k + "_per_launch": v["sum"] / v["launches"] for k, v in raw.items()}
rec = {"rays_per_launch": rays / launches, "launches": launches, "waves_per_simd": 4,
       "node_steps_per_ray": node / rays, "tri_steps_per_ray": tri / rays,
       "node_phase_lanes_active": node / (64.0 * wn), "tri_phase_lanes_active": tri / (64.0 * wt),
       "FETCH_SIZE_KB_per_launch": per.pop("FETCH_SIZE_per_launch"), "WRITE_SIZE_KB_per_launch": per.pop("WRITE_SIZE_per_launch"),
       "source": f"profiles/{tag}_pmc_c5.txt (tools/pmc_c5.sh: rocprofv3 --pmc passes over one full frame, `bench.py --steps 1 --warmup 0 --no-cpu --no-side`; "
                 "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md)"}
rec.update(per)
# the kernels these counters were taken on: bench.py recomputes this hash and refuses to price a newer build with old counters
import hashlib
h = hashlib.sha256()
for f in ("ray-tracing_amd/csrc/rt_kernels.h", "ray-tracing_amd/csrc/rt_device.h", "ray-tracing_amd/csrc/rtgpu.hip"):
    h.update(open(os.path.join(ROOT, f), "rb").read())
rec["kernel_source_sha256"] = h.hexdigest()
path = os.path.join(ROOT, "profiles", "c5_k_shadow_pmc.json")
allrec = json.load(open(path)) if os.path.exists(path) else {}
allrec[key] = rec
json.dump(allrec, open(path, "w"), indent=1)
tcv, act, wc = rec["SQ_THREAD_CYCLES_VALU_per_launch"], rec["SQ_ACTIVE_INST_VALU_per_launch"], rec["SQ_WAVE_CYCLES_per_launch"]
hbm = (2 * rec["FETCH_SIZE_KB_per_launch"] + rec["WRITE_SIZE_KB_per_launch"]) * 1024
with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_c5.txt"), "w") as f:
    f.write(f"rocprofv3 PMC passes over the benched C5 frame ({key}), k_shadow only, {launches} launches, {rays} shadow rays\n")
    for k, v in sorted(raw.items()):
        f.write(f"{k:32s} sum {v['sum']:.6g}  per launch {v['sum'] / v['launches']:.6g}\n")
    f.write(f"RT_COUNT: node steps {node} tri steps {tri} (lane level); wave iterations node {wn} tri {wt}\n")
    f.write(f"=> node steps / ray {node / rays:.2f}, triangle steps / ray {tri / rays:.2f}; lanes active: node phase {node / (64.0 * wn):.3f}, triangle phase {tri / (64.0 * wt):.3f}\n")
    f.write(f"=> VALU lane-ops / ray {tcv / rec['rays_per_launch']:.1f}; lanes active per VALU instruction {tcv / (64 * act):.3f}; VALU busy of the 4 wave slots {4 * act / wc:.3f}\n")
    f.write(f"=> HBM-side bytes / launch {hbm:.4g} ({hbm / rec['rays_per_launch']:.1f} per ray); L2 hit {raw['TCC_HIT_sum']['sum'] / (raw['TCC_HIT_sum']['sum'] + raw['TCC_MISS_sum']['sum']):.3f}; "
            f"L1 hit {1 - raw['TCP_TCC_READ_REQ_sum']['sum'] / raw['TCP_TOTAL_CACHE_ACCESSES_sum']['sum']:.3f}\n")
    f.write(f"bench line of the RT_COUNT run (diagnostic counters cost time): {bench['value']} Mrays/s\n")
print(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_c5.txt")).read())
