#!/bin/bash
# usage: tools/pmc_config.sh <config> <kernel-prefix>  -- SQ and cache counters of one kernel on a side configuration (tools/bench_configs.py)
set -u
cfg=$1; kern=$2
out=$PWD/gpurun_out/pmc_${cfg}_${kern}
mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" -d $out/$name -o p --output-format csv -- python3 $root/tools/bench_configs.py $cfg --steps 1 --warmup 0 > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -n 5 $out/$name.log; return 1; }
  python3 - "$out/$name" "$kern" <<'PY'
import csv, sys, glob, collections
tot = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if n == sys.argv[2]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
print({k: f"{v:.4g}" for k, v in tot.items()})
PY
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD | tee -a $out/summary.txt &&
run tc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum | tee -a $out/summary.txt
