#!/bin/bash
# usage: tools/pmc.sh <tag>   (env passes through) -- two PMC passes over the reduced C5 frame, k_shadow rows only
set -u
tag=$1
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
run() {  # name counters...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" -d $out/$name -o p --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu --width 1920 --height 1080 --spp 4 > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -n 5 $out/$name.log; return 1; }
  python3 - "$out/$name" <<'PY'
import csv, sys, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("void k_shadow<") or r["Kernel_Name"].startswith("k_shadow("):
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
print({k: f"{v:.4g}" for k, v in tot.items()})
PY
}
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS | tee -a $out/summary.txt &&
run tc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum | tee -a $out/summary.txt
