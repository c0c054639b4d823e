// Vector-L1 (TCP) access-rate microbenchmark: divergent per-lane dwordx4 loads of 64-byte records, as k_shadow's node fetch does.
// How many lane-accesses (one lane x one dwordx4) per clock does a CU sustain, from an L1-resident and from an L2-resident table?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int LOADS>   // dwordx4 loads per 64-byte record
__global__ void __launch_bounds__(64) k_l1(const uint4* __restrict__ tab, uint32_t mask, int iters, uint32_t* out) {
    uint32_t idx = (blockIdx.x * 64 + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const uint4* p = tab + (size_t)(idx & mask) * 4;
        uint4 a = p[0];
        acc += a.x;
        if (LOADS > 1) { uint4 b = p[1]; acc += b.y; }
        if (LOADS > 2) { uint4 c = p[2]; acc += c.z; }
        if (LOADS > 3) { uint4 d = p[3]; acc += d.w; }
        idx = idx * 1664525u + 1013904223u + (acc & 1u);      // next record depends on the data: one record in flight per lane, like a tree walk
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

// The same walk with the record fetched by the lane's quad: in step k the four lanes of a quad read the four 16-byte chunks of
// the record that quad lane k wants (one line per quad and instruction), straight into LDS (global_load_lds_dwordx4 lands lane
// L's 16 bytes at base + 16 L, so record (4q+k) lies contiguous at buffer k + 64 q); every lane then reads its own 64 bytes back.
// Buffers are 1040 bytes apart so that the eight lanes a ds_read_b128 serves per clock touch distinct banks.
template <int MODE>   // 0: LDS-DMA, 1: loads to registers + ds_write
__global__ void __launch_bounds__(64) k_coop(const uint4* __restrict__ tab, uint32_t mask, int iters, uint32_t* out) {
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t lane = threadIdx.x;
    uint32_t idx = (blockIdx.x * 64 + lane) * 2654435761u;
    uint32_t acc = 0;
    const uint4* base_lane = tab + (lane & 3);
    const unsigned char* mine = lds + (lane & 3) * 1040 + (lane >> 2) * 64;
    for (int it = 0; it < iters; it++) {
        const uint32_t my = idx & mask;
#define COOP_STEP(k) { const uint32_t ik = __builtin_amdgcn_mov_dpp(my, (k) * 0x55, 0xF, 0xF, true); \
            const uint4* src = base_lane + 4 * (size_t)ik; \
            if (MODE == 0) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(lds + (k) * 1040), 16, 0, 0); \
            else *(uint4*)(lds + (k) * 1040 + lane * 16) = *src; }
        COOP_STEP(0) COOP_STEP(1) COOP_STEP(2) COOP_STEP(3)
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const uint4 a = *(const uint4*)(mine), b = *(const uint4*)(mine + 16), c = *(const uint4*)(mine + 32), d = *(const uint4*)(mine + 48);
        acc += a.x + b.y + c.z + d.w;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        idx = idx * 1664525u + 1013904223u + (acc & 1u);
    }
    out[blockIdx.x * 64 + lane] = acc;
}

template <int MODE>
void run_coop(const uint4* tab, uint32_t records, int waves_per_simd, uint32_t* out) {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, iters = 20000;
    const int blocks = cus * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_coop<MODE><<<blocks, 64, 4160>>>(tab, records - 1, 100, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k_coop<MODE><<<blocks, 64, 4160>>>(tab, records - 1, iters, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double rec_per_cu = (double)iters * 64 * 4 * waves_per_simd;
    printf("table %8.1f KB, quad-cooperative %s, %d waves/SIMD: %.3f ms -> %.2f records (64 B) per 2.4 GHz clock per CU, %.0f ns per dependent step\n",
           records * 64 / 1024.0, MODE == 0 ? "LDS-DMA" : "load + ds_write", waves_per_simd, ms, rec_per_cu / (ms * 1e-3 * 2.4e9), ms * 1e6 / iters);
}

template <int LOADS>
void run(const uint4* tab, uint32_t records, int waves_per_simd, uint32_t* out) {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, iters = 20000;
    const int blocks = cus * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_l1<LOADS><<<blocks, 64>>>(tab, records - 1, 100, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k_l1<LOADS><<<blocks, 64>>>(tab, records - 1, iters, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double acc_per_cu = (double)iters * LOADS * 64 * 4 * waves_per_simd;
    printf("table %8.1f KB, %d dwordx4 per record, %d waves/SIMD: %.3f ms -> %.2f lane-accesses per 2.4 GHz clock per CU, %.0f ns per dependent step\n",
           records * 64 / 1024.0, LOADS, waves_per_simd, ms, acc_per_cu / (ms * 1e-3 * 2.4e9), ms * 1e6 / iters);
}

int main() {
    const uint32_t max_records = 1u << 24;        // 1 GiB
    uint4* tab;
    uint32_t* out;
    CHECK(hipMalloc(&tab, (size_t)max_records * 64));
    CHECK(hipMemset(tab, 0, (size_t)max_records * 64));
    CHECK(hipMalloc(&out, 256 * 4 * 16 * 64 * sizeof(uint32_t)));
    for (uint32_t records : {128u, 1u << 14, 1u << 24})          // 8 KB (L1), 1 MB (L2), 1 GiB (HBM / MALL)
        for (int w : {4, 8}) {
            run<1>(tab, records, w, out);
            run<2>(tab, records, w, out);
            run<4>(tab, records, w, out);
            run_coop<0>(tab, records, w, out);
            run_coop<1>(tab, records, w, out);
        }
    return 0;
}
