#!/bin/bash
# usage: tools/forced_paths.sh -- tests/test_gpu_parity.py once per alternative path forced on for every scene; logs under gpurun_out/forced/
mkdir -p gpurun_out/forced
for v in RT_FRONT=1 RT_FORCE_LEAF_SPLIT=1 RT_NO_LEAF_SPLIT=1 RT_TOP_MIN=48 RT_CBLOCK=256 RT_GROUP_CULL=1 RT_NO_FLAT_TOP=1 RT_WIDE_CLOSEST=1; do
  echo "== $v"
  env $v timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_golden_vectors.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/forced/$v.log 2>&1 || { echo "FAILED under $v"; tail -n 25 gpurun_out/forced/$v.log; exit 1; }
  tail -n 1 gpurun_out/forced/$v.log
done
