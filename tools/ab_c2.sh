#!/bin/bash
# usage: tools/ab_c2.sh libA libB ... ; the default scene (fat leaves, many top-level objects) at 2 spp with each library
mkdir -p gpurun_out
for v in "$@"; do
  echo "== $v" | tee -a gpurun_out/ab_c2.log
  RT_LIB=$PWD/ray-tracing_amd/$v timeout -k 10 300 python -u tools/bench_configs.py C2 --spp ${SPP:-2} --steps 1 --warmup 1 > gpurun_out/abc2_$v.json 2> gpurun_out/abc2_$v.err || { echo "FAILED $v rc=$?" | tee -a gpurun_out/ab_c2.log; tail -n 5 gpurun_out/abc2_$v.err; exit 1; }
  python3 -c "import sys,json; d=json.loads(open('gpurun_out/abc2_$v.json').read().strip().splitlines()[-1]); print(d['ms_per_frame'], d['frame_checksum'], d['kernel_ms_per_frame'])" | tee -a gpurun_out/ab_c2.log
done
