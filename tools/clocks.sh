#!/bin/bash
# samples GPU clocks/power with rocm-smi while the reduced bench runs
mkdir -p gpurun_out
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/clocks.log 2>&1 &
SM=$!
timeout -k 10 240 python -u bench.py --steps 3 --warmup 0 --no-cpu --width 1920 --height 1080 --spp 4 > gpurun_out/clocks_bench.json 2> gpurun_out/clocks_bench.err
wait $SM
sort gpurun_out/clocks.log | uniq -c | sort -rn | head -12
