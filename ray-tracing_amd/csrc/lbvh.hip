/*
 * lbvh.hip -- a mesh tree built on the GPU ("BVH construction on GPU"): a linear BVH in the layout the host
 * builders produce (rt_bvh_node / rt_bvh_leaf / item array, include/rtgpu.h), so everything downstream of the tree --
 * rt_upload_scene's records, hit bounds, quantised 4-wide shadow tree, the kernels -- takes it like any other.
 *
 * NOT the reference's tree (AccelerationStruct.cpp:151-243 splits at the spatial median of the longest axis, depth
 * first, on the CPU): a non-parity mode like the binned-SAH one (Settings::tree_mode = 2), off by default. Which
 * triangles a ray can reach differs from the reference's topology in grazing cases (tests/test_sah_mode.py bounds it).
 *
 * Method (Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", HPG 2012):
 *   1. bounds of the triangle centroids (wave reduction + float atomics on the ordered-integer image of a float);
 *   2. 30-bit Morton code of every centroid, key = code << 32 | triangle index (unique keys: no duplicate handling);
 *   3. rocPRIM radix sort of the keys;
 *   4. one thread per inner node: its key range from the common-prefix lengths of neighbouring keys, the split, the
 *      two child links, the children's parent pointers;
 *   5. one thread per leaf: the triangle's box (its three vertices, as BuildingTriangle does, AccelerationStruct.cpp:36-55)
 *      written into its parent's child slot; the second thread to arrive at a node (atomic counter) goes on upwards
 *      with the union -- every node's two child boxes are complete when the kernel ends.
 * Leaves hold one triangle each (the reference's depth-24 tree over the 10 M-triangle soup holds 1.2).
 */
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string.h>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "rtgpu.h"
#include "lbvh.h"

namespace {

__device__ __forceinline__ uint32_t ordered(float f) {       /* monotone map float -> uint32 */
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float unordered(uint32_t o) {
    const uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    float f;
#ifdef __HIP_DEVICE_COMPILE__
    f = __uint_as_float(b);
#else
    memcpy(&f, &b, 4);
#endif
    return f;
}

__device__ __forceinline__ void centroid(const float* __restrict__ pos, const uint32_t* __restrict__ tri, uint32_t t, float c[3]) {
    const uint32_t a = tri[3 * (size_t)t], b = tri[3 * (size_t)t + 1], d = tri[3 * (size_t)t + 2];
    for (int k = 0; k < 3; k++) c[k] = (pos[3 * (size_t)a + k] + pos[3 * (size_t)b + k] + pos[3 * (size_t)d + k]) / 3.f;   /* AccelerationStruct.cpp:40 */
}

/* 1. centroid bounds: mm[0..2] = min, mm[3..5] = max, as ordered integers */
__global__ void __launch_bounds__(256) k_centroid_bounds(const float* __restrict__ pos, const uint32_t* __restrict__ tri, uint32_t n, uint32_t* mm) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        float c[3];
        centroid(pos, tri, t, c);
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], c[k]); hi[k] = fmaxf(hi[k], c[k]); }
    }
    for (int k = 0; k < 3; k++) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; k++) {
            if (lo[k] <= hi[k]) { atomicMin(mm + k, ordered(lo[k])); atomicMax(mm + 3 + k, ordered(hi[k])); }
        }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v) {    /* 10 bits -> every third bit */
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

/* 2. keys */
__global__ void __launch_bounds__(256) k_morton_keys(const float* __restrict__ pos, const uint32_t* __restrict__ tri, uint32_t n, const uint32_t* __restrict__ mm,
                                                     unsigned long long* keys) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float c[3];
    centroid(pos, tri, t, c);
    uint32_t q[3];
    for (int k = 0; k < 3; k++) {
        const float lo = unordered(mm[k]), hi = unordered(mm[3 + k]);
        const float ext = hi - lo;
        float u = ext > 0.f ? (c[k] - lo) / ext : 0.f;
        u = fminf(fmaxf(u, 0.f), 1.f);
        q[k] = min(1023u, (uint32_t)(u * 1024.f));
    }
    const uint32_t code = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
    keys[t] = ((unsigned long long)code << 32) | t;
}

/* 4. the hierarchy. delta(i, j) = length of the common prefix of keys i and j, -1 outside the array */
__device__ __forceinline__ int delta(const unsigned long long* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}
__global__ void __launch_bounds__(256) k_hierarchy(const unsigned long long* __restrict__ keys, int n, rt_bvh_node* nodes, int* parent /* n-1 inner, then n leaves */) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const bool leaf_l = lo == gamma, leaf_r = hi == gamma + 1;
    nodes[i].left = leaf_l ? -1 - gamma : gamma;
    nodes[i].right = leaf_r ? -1 - (gamma + 1) : gamma + 1;
    nodes[i].pad0[0] = nodes[i].pad0[1] = 0;
    parent[leaf_l ? (n - 1) + gamma : gamma] = i;
    parent[leaf_r ? (n - 1) + gamma + 1 : gamma + 1] = i;
    if (i == 0) parent[0] = -1;
}

/* 5. boxes, bottom-up; items and leaves on the way */
__global__ void __launch_bounds__(256) k_fit(const float* __restrict__ pos, const uint32_t* __restrict__ tri, const unsigned long long* __restrict__ keys, int n,
                                             rt_bvh_node* nodes, const int* __restrict__ parent, int* arrived, rt_bvh_leaf* leaves, uint32_t* items) {
    const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= n) return;
    const uint32_t t = (uint32_t)(keys[leaf] & 0xFFFFFFFFull);
    items[leaf] = t;
    leaves[leaf].first = (uint32_t)leaf;
    leaves[leaf].count = 1u;
    float mn[3], mx[3];
    {
        const uint32_t a = tri[3 * (size_t)t], b = tri[3 * (size_t)t + 1], c = tri[3 * (size_t)t + 2];
        for (int k = 0; k < 3; k++) {
            const float pa = pos[3 * (size_t)a + k], pb = pos[3 * (size_t)b + k], pc = pos[3 * (size_t)c + k];
            mn[k] = fminf(pc, fminf(pb, pa));
            mx[k] = fmaxf(pc, fmaxf(pb, pa));
        }
    }
    int link = -1 - leaf;
    int p = parent[(n - 1) + leaf];
    while (p >= 0) {
        rt_bvh_node* nd = nodes + p;
        const bool is_left = nd->left == link;
        float* dmn = is_left ? nd->lmin : nd->rmin;
        float* dmx = is_left ? nd->lmax : nd->rmax;
        for (int k = 0; k < 3; k++) { dmn[k] = mn[k]; dmx[k] = mx[k]; }
        __threadfence();
        if (atomicAdd(arrived + p, 1) == 0) return;           /* the sibling's thread will carry the union upwards */
        __threadfence();
        const volatile float* omn = is_left ? nd->rmin : nd->lmin;
        const volatile float* omx = is_left ? nd->rmax : nd->lmax;
        for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], omn[k]); mx[k] = fmaxf(mx[k], omx[k]); }
        link = p;
        p = parent[p];
    }
}

struct Buf {
    void* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 16)); }
    template <class T> T* as() { return static_cast<T*>(p); }
};

}  // namespace

#define LBVH_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(err, err_len, "%s: %s", #x, hipGetErrorString(e_)); return RT_ERR_DEVICE; } } while (0)

extern "C" int rtgpu_lbvh_build(const float* positions, int64_t n_verts, const uint32_t* tris, int64_t n_tris, rt_bvh_node* nodes_out,
                                rt_bvh_leaf* leaves_out, uint32_t* items_out, int32_t* root_out, int32_t* depth_out, double* device_seconds,
                                char* err, size_t err_len) {
    if (err && err_len) err[0] = 0;
    if (!positions || !tris || !nodes_out || !leaves_out || !items_out || !root_out || !depth_out || n_verts < 1 || n_tris < 2 || n_tris > 0x0FFFFFF0ll) {
        snprintf(err, err_len, "rtgpu_lbvh_build: bad arguments");
        return RT_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { snprintf(err, err_len, "the GPU tree builder needs a GPU: none visible"); return RT_ERR_DEVICE; }
    const int n = (int)n_tris;
    for (int64_t i = 0; i < 3 * n_tris; i++)
        if (tris[i] >= (uint64_t)n_verts) { snprintf(err, err_len, "triangle index out of range"); return RT_ERR_ARG; }
    Buf d_pos, d_tri, d_mm, d_keys, d_keys2, d_tmp, d_nodes, d_parent, d_arrived, d_leaves, d_items;
    LBVH_TRY(d_pos.alloc((size_t)n_verts * 12));
    LBVH_TRY(d_tri.alloc((size_t)n * 12));
    LBVH_TRY(d_mm.alloc(6 * 4));
    LBVH_TRY(d_keys.alloc((size_t)n * 8));
    LBVH_TRY(d_keys2.alloc((size_t)n * 8));
    LBVH_TRY(d_nodes.alloc((size_t)(n - 1) * sizeof(rt_bvh_node)));
    LBVH_TRY(d_parent.alloc((size_t)(2 * n - 1) * 4));
    LBVH_TRY(d_arrived.alloc((size_t)(n - 1) * 4));
    LBVH_TRY(d_leaves.alloc((size_t)n * sizeof(rt_bvh_leaf)));
    LBVH_TRY(d_items.alloc((size_t)n * 4));
    LBVH_TRY(hipMemcpy(d_pos.p, positions, (size_t)n_verts * 12, hipMemcpyHostToDevice));
    LBVH_TRY(hipMemcpy(d_tri.p, tris, (size_t)n * 12, hipMemcpyHostToDevice));
    const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    LBVH_TRY(hipMemcpy(d_mm.p, init, sizeof init, hipMemcpyHostToDevice));
    LBVH_TRY(hipMemset(d_arrived.p, 0, (size_t)(n - 1) * 4));
    size_t tmp_bytes = 0;
    LBVH_TRY(rocprim::radix_sort_keys(nullptr, tmp_bytes, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(), (size_t)n, 0, 62));
    LBVH_TRY(d_tmp.alloc(tmp_bytes));
    struct Events {
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    LBVH_TRY(hipEventCreate(&ev.a));
    LBVH_TRY(hipEventCreate(&ev.b));
    hipEvent_t e0 = ev.a, e1 = ev.b;
    LBVH_TRY(hipEventRecord(e0, 0));
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(k_centroid_bounds, dim3(std::min<uint32_t>(blocks, 4096u)), dim3(256), 0, 0, d_pos.as<float>(), d_tri.as<uint32_t>(), (uint32_t)n, d_mm.as<uint32_t>());
    hipLaunchKernelGGL(k_morton_keys, dim3(blocks), dim3(256), 0, 0, d_pos.as<float>(), d_tri.as<uint32_t>(), (uint32_t)n, d_mm.as<uint32_t>(), d_keys.as<unsigned long long>());
    LBVH_TRY(rocprim::radix_sort_keys(d_tmp.p, tmp_bytes, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(), (size_t)n, 0, 62));
    hipLaunchKernelGGL(k_hierarchy, dim3(blocks), dim3(256), 0, 0, d_keys2.as<unsigned long long>(), n, d_nodes.as<rt_bvh_node>(), d_parent.as<int>());
    hipLaunchKernelGGL(k_fit, dim3(blocks), dim3(256), 0, 0, d_pos.as<float>(), d_tri.as<uint32_t>(), d_keys2.as<unsigned long long>(), n, d_nodes.as<rt_bvh_node>(),
                       d_parent.as<int>(), d_arrived.as<int>(), d_leaves.as<rt_bvh_leaf>(), d_items.as<uint32_t>());
    LBVH_TRY(hipEventRecord(e1, 0));
    LBVH_TRY(hipEventSynchronize(e1));
    LBVH_TRY(hipGetLastError());
    float ms = 0.f;
    LBVH_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (device_seconds) *device_seconds = ms * 1e-3;
    LBVH_TRY(hipMemcpy(nodes_out, d_nodes.p, (size_t)(n - 1) * sizeof(rt_bvh_node), hipMemcpyDeviceToHost));
    LBVH_TRY(hipMemcpy(leaves_out, d_leaves.p, (size_t)n * sizeof(rt_bvh_leaf), hipMemcpyDeviceToHost));
    LBVH_TRY(hipMemcpy(items_out, d_items.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    *root_out = 0;
    /* depth (levels of inner nodes above the deepest leaf), for the stack capacities of the walks */
    int depth = 0;
    {
        std::vector<std::pair<int32_t, int32_t>> todo;
        todo.emplace_back(0, 1);
        while (!todo.empty()) {
            const auto [node, level] = todo.back();
            todo.pop_back();
            depth = std::max(depth, level);
            if (nodes_out[node].left >= 0) todo.emplace_back(nodes_out[node].left, level + 1);
            if (nodes_out[node].right >= 0) todo.emplace_back(nodes_out[node].right, level + 1);
        }
    }
    *depth_out = depth;
    if (depth > RT_MAX_TREE_DEPTH) { snprintf(err, err_len, "the GPU-built tree is %d levels deep (limit %d)", depth, RT_MAX_TREE_DEPTH); return RT_ERR_LIMIT; }
    return RT_OK;
}
