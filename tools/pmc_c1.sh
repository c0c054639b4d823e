#!/bin/bash
# usage: tools/pmc_c1.sh <tag>  -- PMC passes over one frame of C1 (default scene, 480x480, 30 spp, depth 50), k_closest rows only
# (the bounce path: primaries + reflection / refraction states). Raw sums -> gpurun_out/<tag>/pmc_*.json;
# tools/pmc_c1_json.py turns them into profiles/c1_k_closest_pmc.json (what bench.py reads for side_configs.C1) and a text summary.
set -u
tag=$1; shift
out=$PWD/gpurun_out/$tag
mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
run() {  # name counters...
  local name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" -d $out/$name -o p --output-format csv -- python3 $root/tools/bench_configs.py C1 --steps 1 --warmup 0 > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -n 5 $out/$name.log; return 1; }
  python3 - "$out/$name" "$out/pmc_$name.json" <<'PY'
import csv, sys, glob, collections, json
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("void k_closest<"):
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
res = {k: {"sum": v, "launches": n[k]} for k, v in tot.items()}
json.dump(res, open(sys.argv[2], "w"), indent=1)
print({k: f"{v:.5g} over {n[k]}" for k, v in tot.items()})
PY
  rm -rf $out/$name
}
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS GRBM_GUI_ACTIVE &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run tc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum &&
timeout -k 10 200 python3 -u $root/tools/bench_configs.py C1 --steps 3 --warmup 1 > $out/plain.json 2> $out/plain.err && tail -n 1 $out/plain.json
