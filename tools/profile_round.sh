#!/bin/bash
# usage: tools/profile_round.sh <tag>  -- default bench line, rocprofv3 kernel stats of the same command, HBM traffic passes.
# Writes everything under gpurun_out/<tag>/ ; copy the summaries into profiles/ afterwards.
set -u
tag=$1
root=$PWD
out=$root/gpurun_out/$tag
mkdir -p $out
timeout -k 10 600 python3 -u bench.py > $out/bench.json 2> $out/bench.err || { echo "bench failed"; tail -n 5 $out/bench.err; exit 1; }
tail -n 1 $out/bench.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 $root/bench.py --no-cpu > $out/trace.log 2>&1 || { echo "trace failed"; tail -n 5 $out/trace.log; exit 1; }
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
rm -f $(find $out/trace -name "*kernel_trace.csv")
head -n 8 $out/kernel_stats.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $ctr -d $out/$ctr -o p --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu > $out/$ctr.log 2>&1 || { echo "$ctr pass failed"; tail -n 5 $out/$ctr.log; exit 1; }
  python3 - "$out/$ctr" $ctr <<'PY' | tee -a $out/pmc_summary.txt
import csv, sys, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        tot[k] += float(r["Counter_Value"]); n[k] += 1
for k in sorted(tot, key=lambda k: -tot[k])[:5]:
    print(sys.argv[2], k, f"total {tot[k]:.5g} KB over {n[k]} launches = {tot[k]/n[k]:.5g} KB/launch")
PY
  rm -rf $out/$ctr
done
