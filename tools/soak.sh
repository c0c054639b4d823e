#!/bin/bash
# usage: tools/soak.sh A B -- test_random_scenes (hits, shadow queries, radiance of random free-form scenes against the oracle) on seeds A..B-1; log under gpurun_out/soak/
mkdir -p gpurun_out/soak
RT_SOAK_SEEDS=$1:$2 timeout -k 10 1000 python3 -u -m pytest tests/test_gpu_parity.py -m gpu -q -k test_random_scenes --timeout 120 --timeout-method thread > gpurun_out/soak/$1_$2.log 2>&1
rc=$?
tail -n 15 gpurun_out/soak/$1_$2.log
exit $rc
