#!/bin/bash
# usage: tools/count_phases.sh -- RT_COUNT=1 step and wave-clock counters of k_shadow on the reduced C5 frame, C2, the gorilla and the cow; logs under gpurun_out/count/
mkdir -p gpurun_out/count
RT_COUNT=1 timeout -k 10 300 python -u bench.py --steps 1 --warmup 0 --no-cpu --no-side --width 1920 --height 1080 --spp 4 > gpurun_out/count/c5.json 2> gpurun_out/count/c5.err || exit 1
for cfg in C2 C4 C4cow chamber; do
  RT_COUNT=1 timeout -k 10 300 python -u tools/bench_configs.py $cfg --steps 1 --warmup 0 > gpurun_out/count/$cfg.json 2> gpurun_out/count/$cfg.err || exit 1
done
grep -h "shadow steps\|wave clocks" gpurun_out/count/*.err | sort | uniq -c | sort -rn | head -5
