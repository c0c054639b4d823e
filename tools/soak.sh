#!/bin/bash
# usage: tools/soak.sh A B [frames] -- test_random_scenes (hits, shadow queries, radiance of random free-form scenes against the oracle) on seeds
# A..B-1, or with a third argument `frames` test_random_scene_frames (the frame path on the same scenes), with `wide` the second family of scenes, with `hall` the mirror halls at depth 50, with `grid` the height fields with rays through shared vertices and edges; log under gpurun_out/soak/
mkdir -p gpurun_out/soak
if [ "$3" = hall ]; then var=RT_SOAK_HALL_SEEDS; sel=halls_of_mirrors; elif [ "$3" = grid ]; then var=RT_SOAK_GRID_SEEDS; sel=shared_vertices; elif [ "$3" = wide ]; then var=RT_SOAK_WIDE_SEEDS; sel=second_family; elif [ -n "$3" ]; then var=RT_SOAK_FRAME_SEEDS; sel=test_random_scene_frames; else var=RT_SOAK_SEEDS; sel="test_random_scenes"; fi
env $var=$1:$2 timeout -k 10 1100 python3 -u -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "$sel" --timeout 150 --timeout-method thread > gpurun_out/soak/$1_$2$3.log 2>&1
rc=$?
tail -n 25 gpurun_out/soak/$1_$2$3.log
exit $rc
