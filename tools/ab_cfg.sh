#!/bin/bash
# usage: tools/ab_cfg.sh <config> libA libB ... ; a side configuration (tools/bench_configs.py) with each library
cfg=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  echo "== $cfg $v" | tee -a gpurun_out/ab_cfg.log
  RT_LIB=$PWD/ray-tracing_amd/$v timeout -k 10 300 python -u tools/bench_configs.py $cfg --steps 2 --warmup 1 ${SPP:+--spp $SPP} > gpurun_out/abcfg_$v.json 2> gpurun_out/abcfg_$v.err || { echo "FAILED $v rc=$?" | tee -a gpurun_out/ab_cfg.log; tail -n 5 gpurun_out/abcfg_$v.err; exit 1; }
  python3 -c "import sys,json; d=json.loads(open('gpurun_out/abcfg_$v.json').read().strip().splitlines()[-1]); print(d['ms_per_frame'], d['frame_checksum'], d['kernel_ms_per_frame'])" | tee -a gpurun_out/ab_cfg.log
done
