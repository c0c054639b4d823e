#!/bin/bash
# usage: tools/ab_c4.sh libA libB ... ; gorilla, cow and chamber frames with each library (RT_LIB)
for v in "$@"; do
  echo "== $v"
  export RT_LIB=$PWD/ray-tracing_amd/$v
  for c in C4 C4cow chamber; do timeout -k 10 100 python tools/bench_configs.py $c --steps 3 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['ms_per_frame'], d['kernel_ms_per_frame'], d['frame_checksum'])" || exit 1; done
done
