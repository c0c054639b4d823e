/* lbvh.h -- the GPU tree builder (lbvh.hip) as the host mirror (host/scene.cpp, plain C++) sees it. Internal to librtgpu.so. */
#pragma once
#include <cstddef>
#include <cstdint>
#include "rtgpu.h"

/* A linear BVH over the n_tris triangles of one mesh, built on the current HIP device, in the layout of the host builders:
 * nodes_out[n_tris - 1], leaves_out[n_tris] (one triangle each), items_out[n_tris] (triangle indices in leaf order),
 * root link 0, depth as rth::FlatTree counts it. n_tris >= 2. Returns RT_OK or a negative rt_status with a text in err. */
extern "C" int rtgpu_lbvh_build(const float* positions, int64_t n_verts, const uint32_t* tris, int64_t n_tris, rt_bvh_node* nodes_out,
                                rt_bvh_leaf* leaves_out, uint32_t* items_out, int32_t* root_out, int32_t* depth_out, double* device_seconds,
                                char* err, size_t err_len);
