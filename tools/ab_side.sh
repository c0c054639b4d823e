#!/bin/bash
for v in "$@"; do
  echo "== $v"
  export RT_LIB=$PWD/ray-tracing_amd/$v
  for c in C2 C4 C4cow chamber; do timeout -k 10 100 python tools/bench_configs.py $c --steps 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['ms_per_frame'], d['kernel_ms_per_frame']['k_shadow'], d['frame_checksum'])" || exit 1; done
done
