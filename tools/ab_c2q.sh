#!/bin/bash
# usage: tools/ab_c2q.sh libA libB ... ; the default scene at 16 spp with each library, 60 s limit per run (experiments on loop conditions)
for v in "$@"; do
  RT_LIB=$PWD/ray-tracing_amd/$v timeout -k 5 60 python -u tools/bench_configs.py C2 --steps 3 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_frame'], d['frame_checksum'], d['kernel_ms_per_frame'])" || { echo "FAILED $v"; exit 1; }
done
