#!/bin/bash
# usage: tools/pmc_kernels.sh <tag> <config> [bench_configs args]  -- one SQ counter pass over a side config (tools/bench_configs.py, one frame,
# no warm-up), summed per kernel: wave instructions, lane-ops, lanes active, VALU busy share. Diagnostic; prints a table.
set -u
tag=$1; cfg=$2; shift; shift
out=$PWD/gpurun_out/$tag
mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $out/sq -o p --output-format csv -- python3 $root/tools/bench_configs.py $cfg --steps 1 --warmup 0 "$@" > $out/sq.log 2>&1 || { echo "pass failed"; tail -n 5 $out/sq.log; exit 1; }
python3 - "$out/sq" <<'PY'
import csv, sys, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
print(f"{'kernel':40s} {'calls':>5s} {'VALU insts':>11s} {'lane-ops':>11s} {'lanes act':>9s} {'VALU busy':>9s} {'wait':>6s} {'SALU/VALU':>9s} {'VMEM rd':>10s}")
for k in sorted(tot, key=lambda k: -tot[k]["SQ_WAVE_CYCLES"])[:10]:
    t = tot[k]
    iv = max(t["SQ_INSTS_VALU"], 1.0); wc = max(t["SQ_WAVE_CYCLES"], 1.0)
    print(f"{k:40s} {n[k]:5d} {iv:11.4g} {t['SQ_THREAD_CYCLES_VALU']:11.4g} {t['SQ_THREAD_CYCLES_VALU'] / (64 * iv):9.3f} {t['SQ_ACTIVE_INST_VALU'] / wc:9.3f} {t['SQ_WAIT_ANY'] / wc:6.3f} {t['SQ_INSTS_SALU'] / iv:9.3f} {t['SQ_INSTS_VMEM_RD']:10.4g}")
PY
rm -rf $out/sq
grep '^{' $out/sq.log | cut -c1-400
