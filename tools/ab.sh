#!/bin/bash
# usage: scratch/ab2.sh libA libB ... ; runs the reduced config with each library
mkdir -p gpurun_out
for v in "$@"; do
  echo "== $v" | tee -a gpurun_out/ab.log
  RT_LIB=$PWD/ray-tracing_amd/$v timeout -k 10 240 python -u bench.py --steps 1 --warmup 0 --no-cpu --width 1920 --height 1080 --spp 4 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "FAILED $v rc=$?" | tee -a gpurun_out/ab.log; tail -n 5 gpurun_out/ab_$v.err; exit 1; }
  python3 -c "import sys,json; d=json.loads(open('gpurun_out/ab_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['frame_checksum_matches_reference_order_traversal'], d['config']['kernel_ms_per_step'])" | tee -a gpurun_out/ab.log
done
