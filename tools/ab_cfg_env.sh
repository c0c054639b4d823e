#!/bin/bash
# usage: tools/ab_env.sh <config> VAR v1 v2 ... ; the config's frame time with the environment variable VAR set to each value
cfg=$1; var=$2; shift; shift
for v in "$@"; do
  env $var=$v timeout -k 10 200 python -u tools/bench_configs.py $cfg ${SPP:+--spp $SPP} --steps 3 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg $var=$v', d['ms_per_frame'], d['frame_checksum'], d['kernel_ms_per_frame'], d['rays_per_frame']['secondary'])" || exit 1
done
