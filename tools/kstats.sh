#!/bin/bash
# usage: tools/kstats.sh <config> [bench_configs args]  -- rocprofv3 kernel stats of one side config (tools/bench_configs.py)
set -u
cfg=$1; shift
root=$PWD
out=$root/gpurun_out/kstats_$cfg
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 $root/tools/bench_configs.py $cfg "$@" > $out/run.log 2>&1 || { echo "trace failed"; tail -n 5 $out/run.log; exit 1; }
find $out/trace -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/trace
grep '^{' $out/run.log | cut -c1-300
python3 - $out/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {float(r["Percentage"]):5.1f}% {int(r["Calls"]):5d} calls  {r["Name"][:70]}')
print(f"total {tot/1e6:.2f} ms over all launches of the run (warm-up + timed frames)")
PY
