/* tools/emu/hip_shim.h -- just enough of the HIP names for g++ to compile rt_device.h / rt_kernels.h as plain C++ (RT_HOST_EMU).
 * One lane: ballots see one lane, shuffles return their argument, atomics are plain operations. The kernels compile but are never
 * called; the emulator drives the per-lane functions of the shadow walk (shadow_emu.cpp). Test infrastructure only. */
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__
#define __align__(x) alignas(x)
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct dim3 { unsigned x = 1, y = 1, z = 1; };
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { uint2 r; r.x = x; r.y = y; return r; }
static dim3 threadIdx{0, 0, 0}, blockIdx{0, 0, 0}, blockDim{64, 1, 1}, gridDim{1, 1, 1};
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __lane_id() { return 0; }
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool b) { return b ? 1ull : 0ull; }
static inline unsigned long long __ballot(bool b) { return b ? 1ull : 0ull; }
static inline void __syncthreads() {}
static inline void __threadfence_system() {}
template <class T> static inline T __shfl(T v, int) { return v; }
template <class T> static inline T __shfl_up(T v, int) { return v; }
template <class T> static inline T __shfl_down(T v, int) { return v; }
static inline long long __double2ll_rn(double x) { return llrint(x); }
static inline float __fsqrt_rn(float x) { return sqrtf(x); }
static inline unsigned long long emu_cycles() { return 0ull; }
#define __builtin_readcyclecounter emu_cycles
/* v_perm_b32: byte i of the result is picked by selector byte i from {S0 (bytes 4-7), S1 (bytes 0-3)}; 0x0C gives 0x00, 0x0D and up 0xFF */
static inline uint32_t __builtin_amdgcn_perm(uint32_t s0, uint32_t s1, uint32_t sel) {
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        const uint32_t k = (sel >> (8 * i)) & 0xFFu;
        uint32_t b;
        if (k < 4) b = (s1 >> (8 * k)) & 0xFFu;
        else if (k < 8) b = (s0 >> (8 * (k - 4))) & 0xFFu;
        else if (k == 0x0C) b = 0;
        else if (k >= 0x0D) b = 0xFF;
        else b = 0;      /* (8-11: sign replication; not used here) */
        r |= b << (8 * i);
    }
    return r;
}
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline uint32_t atomicOr(uint32_t* p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
