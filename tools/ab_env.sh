#!/bin/bash
# usage: tools/ab_env.sh "ENV=.. ENV2=.." "..." ; runs the reduced C5 frame (1080p, 4 spp) once per environment string
mkdir -p gpurun_out
for v in "$@"; do
  echo "== $v"
  env $v timeout -k 10 240 python -u bench.py --steps 1 --warmup 0 --no-cpu --no-side --width 1920 --height 1080 --spp 4 > gpurun_out/abenv.json 2> gpurun_out/abenv.err || { echo "FAILED rc=$?"; tail -n 5 gpurun_out/abenv.err; exit 1; }
  grep rtgpu gpurun_out/abenv.err | tail -3
  python3 -c "import sys,json; d=json.loads(open('gpurun_out/abenv.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['frame_checksum_matches_reference_order_traversal'], d['config']['kernel_ms_per_step'])"
done
