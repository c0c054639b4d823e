#!/bin/bash
# usage: tools/soak_variants.sh A B -- the frame soak (seeds A..B-1) and a dozen scenes of the second family under each alternative path forced on
for v in RT_WIDE_LDS_CAP=7 RT_FRONT=1 RT_FRONT=0 RT_SHADOW_EXACT_NODES=1 RT_FORCE_LEAF_SPLIT=1 RT_NO_LEAF_SPLIT=1 RT_WIDE_CLOSEST=1 RT_CBLOCK=256 RT_NO_TOP_SPLIT=1 RT_NO_FLAT_TOP=1 RT_TOP_MIN=48 RT_SEPARATE_RAYGEN=1 RT_SPREAD_LANES=0; do
  echo "== $v"
  env $v bash tools/soak.sh $1 $2 frames 2>&1 | grep -E "passed|failed|FAILED" || exit 1
  env $v bash tools/soak.sh 2000 2012 wide 2>&1 | grep -E "passed|failed|FAILED" || exit 1
done
