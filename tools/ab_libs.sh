#!/bin/bash
# usage: tools/ab_libs.sh lib1.so lib2.so ... ; reduced C5 frame + C2 + C4 per library (RT_LIB)
for v in "$@"; do
  echo "== $v"
  export RT_LIB=$PWD/ray-tracing_amd/$v
  timeout -k 10 240 python -u bench.py --steps 1 --warmup 0 --no-cpu --no-side --width 1920 --height 1080 --spp 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5r', d['value'], d['config']['frame_checksum_matches_reference_order_traversal'], d['config']['kernel_ms_per_step'])" || exit 1
  timeout -k 10 100 python tools/bench_configs.py C2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 ', d['ms_per_frame'], d['kernel_ms_per_frame'], d['frame_checksum'])" || exit 1
  timeout -k 10 100 python tools/bench_configs.py C4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 ', d['ms_per_frame'], d['kernel_ms_per_frame'], d['frame_checksum'])" || exit 1
  for c in C4cow chamber C3; do timeout -k 10 100 python tools/bench_configs.py $c 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['ms_per_frame'], d['kernel_ms_per_frame'], d['frame_checksum'])" || exit 1; done
done
