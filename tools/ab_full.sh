#!/bin/bash
# usage: tools/ab_full.sh "ENV=.." ... ; the full benched C5 frame once per environment string, then C2 / chamber / C3
mkdir -p gpurun_out
for v in "$@"; do
  echo "== $v"
  env $v timeout -k 10 300 python -u bench.py --steps 1 --warmup 0 --no-cpu --no-side > gpurun_out/abfull.json 2> gpurun_out/abfull.err || { echo "FAILED rc=$?"; tail -n 5 gpurun_out/abfull.err; exit 1; }
  python3 -c "import sys,json; d=json.loads(open('gpurun_out/abfull.json').read().strip().splitlines()[-1]); print('C5', d['value'], d['config']['frame_checksum_matches_reference_order_traversal'], d['config']['kernel_ms_per_step'])"
  for c in C2 chamber C3 C4; do env $v timeout -k 10 100 python tools/bench_configs.py $c 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['ms_per_frame'], d['kernel_ms_per_frame'], d['frame_checksum'])" || exit 1; done
done
