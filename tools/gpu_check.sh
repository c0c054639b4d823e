#!/bin/bash
# usage: tools/gpu_check.sh <tag> -- the GPU test suite, the not-yet-collected tests named on the command line, a short bench; logs under gpurun_out/<tag>/
set -u
tag=$1; shift
out=$PWD/gpurun_out/$tag
mkdir -p $out
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread "$@" > $out/pytest.log 2>&1 || { echo "pytest failed"; tail -n 30 $out/pytest.log; exit 1; }
tail -n 3 $out/pytest.log
timeout -k 10 400 python3 -u bench.py --steps 2 --warmup 1 > $out/bench.json 2> $out/bench.err || { echo "bench failed"; tail -n 20 $out/bench.err; exit 1; }
tail -n 1 $out/bench.json
