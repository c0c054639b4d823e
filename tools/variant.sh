#!/bin/bash
# usage: tools/variant.sh <name> [-DRT_X=..]...  -- builds ray-tracing_amd/librtgpu_<name>.so with extra defines (A/B runs via RT_LIB)
set -e
name=$1; shift
cd "$(dirname "$0")/../ray-tracing_amd/csrc"
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -I../../include -I. "$@" -c rtgpu.hip -o build/rtgpu_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../librtgpu_$name.so build/rtgpu_$name.o build/scene.o build/lbvh.o -lz
echo built librtgpu_$name.so
