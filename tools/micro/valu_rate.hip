// VALU issue-rate microbenchmark: how many wave64 VALU instructions per cycle does one SIMD sustain?
// Prices the "valu" roofline peak of bench.py (256 CUs x 4 SIMDs x LANES x clock).  hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int KIND>
__global__ void __launch_bounds__(256) k_rate(float* out, int iters, float a, float b) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : );
                if (KIND == 5) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(x[i]));
                if (KIND == 6) asm volatile("v_cvt_f32_ubyte2 %0, %0" : "+v"(x[i]));
                if (KIND == 7) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 9) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(x[i]), "v"(a) : "vcc");
                if (KIND == 10) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 11) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(x[i]) : "v"(a));
                if (KIND == 12) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 13) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x[i]));
                if (KIND == 14) asm volatile("v_and_b32 %0, 0xffff, %0" : "+v"(x[i]));
                if (KIND == 15) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 16) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 17) asm volatile("v_min3_f32 %0, %0, %1, 0" : "+v"(x[i]) : "v"(a));
                if (KIND == 18) asm volatile("v_cmp_le_f32 s[10:11], %0, %1" : : "v"(x[i]), "v"(a) : "s10", "s11");
                if (KIND == 19) asm volatile("v_cndmask_b32 %0, %0, %1, s[10:11]" : "+v"(x[i]) : "v"(a));
                if (KIND == 20) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 21) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 22) asm volatile("v_bfe_u32 %0, %0, 16, 16" : "+v"(x[i]));
                if (KIND == 23) asm volatile("v_lshrrev_b32 %0, 16, %0" : "+v"(x[i]));
                if (KIND == 24) asm volatile("v_lshl_or_b32 %0, %0, 7, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 25) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 26) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 27) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 28) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 29) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 30) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 31) asm volatile("v_fma_mix_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 32) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 33) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(a));
                if (KIND == 34) asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 35) asm volatile("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 36) asm volatile("v_mad_u32_u16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 37) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 38) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (KIND == 39) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 40) asm volatile("v_cmp_le_i32 s[10:11], %0, %1" : : "v"(x[i]), "v"(a) : "s10", "s11");
                if (KIND == 41) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float pk2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void __launch_bounds__(256) k_rate_pk(float* out, int iters, float a, float b) {
    pk2 x[16];
    pk2 a2 = {a, a + 1.f}, b2 = {b, b + 1.f};
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = pk2{(float)(threadIdx.x + i), (float)i};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a2), "v"(b2));
                if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a2));
                if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a2));
                if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(x[i]) : "v"(a2), "v"(b2));
                if (KIND == 4) asm volatile("v_pk_mov_b32 %0, %1, %2" : "+v"(x[i]) : "v"(a2), "v"(b2));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND, bool PK = false>
void run(const char* name, int blocks_per_cu, float* d_out) {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, iters = 4000;
    const int blocks = cus * blocks_per_cu;                      // 256 threads = 4 waves = one wave per SIMD per block
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    if (PK) k_rate_pk<KIND><<<blocks, 256>>>(d_out, 10, 1.0001f, 0.5f); else k_rate<KIND><<<blocks, 256>>>(d_out, 10, 1.0001f, 0.5f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    if (PK) k_rate_pk<KIND><<<blocks, 256>>>(d_out, iters, 1.0001f, 0.5f); else k_rate<KIND><<<blocks, 256>>>(d_out, iters, 1.0001f, 0.5f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double inst_per_simd = (double)iters * 128 * blocks_per_cu;      // wave-instructions issued on each SIMD
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-14s %d waves/SIMD: %.3f ms, %.2f cycles (at 2.4 GHz) per wave64 instruction per SIMD -> %.1f Tlane-op/s on %d CUs\n", name, blocks_per_cu, ms,
           cyc / inst_per_simd, inst_per_simd * 64 * cus * 4 / (ms * 1e-3) / 1e12, cus);
}

int main() {
    float* d;
    CHECK(hipMalloc(&d, 256 * 256 * 8 * 256 * sizeof(float)));
    for (int w : {1, 4}) {
        run<0>("v_fma_f32", w, d);
        run<1>("v_max3_f32", w, d);
        run<2>("v_perm_b32", w, d);
        run<3>("v_add_u32", w, d);
        run<4>("v_cndmask vcc", w, d);
        run<5>("cvt_u32 sdwa w1", w, d);
        run<6>("cvt_ubyte2", w, d);
        run<7>("v_max_f32", w, d);
        run<8>("v_mul_f32", w, d);
        run<9>("v_cmp_le vcc", w, d);
        run<10>("v_lshl_add_u32", w, d);
        run<11>("v_alignbit", w, d);
        run<12>("v_fmac_f32", w, d);
        run<13>("v_cvt_f32_u32", w, d);
        run<14>("v_and imm", w, d);
        run<15>("fma x,a,a", w, d);
        run<16>("perm x,a,a", w, d);
        run<17>("min3 x,a,0", w, d);
        run<18>("v_cmp_le sgpr", w, d);
        run<19>("cndmask sgpr", w, d);
        run<20>("v_bfi_b32", w, d);
        run<21>("v_and_or_b32", w, d);
        run<22>("v_bfe_u32", w, d);
        run<23>("v_lshrrev_b32", w, d);
        run<24>("v_lshl_or_b32", w, d);
        run<25>("v_min_u32", w, d);
        run<26>("v_med3_f32", w, d);
        run<27>("v_add_f32", w, d);
        run<28>("v_or_b32", w, d);
        run<29>("v_mad_u32_u24", w, d);
        run<30>("v_add3_u32", w, d);
        run<31>("v_fma_mix_f32", w, d);
        run<32>("v_mov_b32", w, d);
        run<33>("v_mov_dpp quad", w, d);
        run<34>("v_mad_i32_i16", w, d);
        run<35>("mad_i32_i16 hi", w, d);
        run<36>("mad_u32_u16 hi", w, d);
        run<37>("v_max3_i32", w, d);
        run<38>("v_max_i32", w, d);
        run<39>("v_dot2_i32_i16", w, d);
        run<40>("v_cmp_le_i32 s", w, d);
        run<41>("v_sub_u32", w, d);
        run<0, true>("v_pk_fma_f32", w, d);
        run<1, true>("v_pk_mul_f32", w, d);
        run<2, true>("v_pk_add_f32", w, d);
        run<3, true>("pk_fma bcast", w, d);
        run<4, true>("v_pk_mov_b32", w, d);
    }
    return 0;
}
