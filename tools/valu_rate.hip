// Micro-benchmark: issue cost of the vector instructions the decode kernels are made of, on gfx950, at full
// occupancy (8 waves per SIMD), eight independent dependency chains per wave, instructions pinned by inline asm.
// Prints cycles per wave-instruction per SIMD at the nominal 2.4 GHz.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float float2v __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float x[8];
    float2v p[8];
    unsigned u[8];
    for (int i = 0; i < 8; i++) {
        x[i] = a + i + threadIdx.x;
        p[i] = (float2v){a + i + threadIdx.x, b + i};
        u[i] = threadIdx.x * 2654435761u + i;
    }
    float2v m = {a, b};
    unsigned um = __float_as_uint(a) | 1u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#define ONE(i)                                                                                                     \
    if (MODE == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));                                     \
    else if (MODE == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));                                \
    else if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m));                             \
    else if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m));                             \
    else if (MODE == 4) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(um));                               \
    else if (MODE == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(um));                               \
    else if (MODE == 6) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(um));                       \
    else if (MODE == 7) asm volatile("v_ffbl_b32 %0, %0" : "+v"(u[i]));                                            \
    else if (MODE == 8) asm volatile("v_mov_b32 %0, %1" : "+v"(u[i]) : "v"(um));                                   \
    else if (MODE == 9) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(u[i]));                                         \
    else if (MODE == 10) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[i]) : "v"(a));                      \
    else if (MODE == 11) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u[i]));                                     \
    else if (MODE == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(um));                     \
    else if (MODE == 13) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(u[i]), "v"(um) : "vcc");                  \
    else if (MODE == 14) asm volatile("v_trunc_f32 %0, %0" : "+v"(x[i]));                                          \
    else if (MODE == 15) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a));                           \
    else if (MODE == 16) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a));                          \
    else if (MODE == 17) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(u[i]));                                      \
    else if (MODE == 18) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(u[i]));                                     \
    else if (MODE == 19) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(um));                           \
    else if (MODE == 20) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(um));                      \
    else if (MODE == 21) asm volatile("v_alignbit_b32 %0, %0, %1, 5" : "+v"(u[i]) : "v"(um));                      \
    else if (MODE == 22) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(m));                        \
    else if (MODE == 23) asm volatile("v_mul_f32 %0, %0, %1\n\tv_and_b32 %2, %2, %3" : "+v"(x[i]), "+v"(u[i]) : "v"(a), "v"(um));
            REP8(ONE)
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += x[i] + p[i].x + p[i].y + __uint_as_float(u[i]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> static void run(float *d, const char *name, int per)
{
    const int iters = 1000, grid = 256 * 8; // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.9999f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double insts_per_simd = 8.0 * iters * 32 * per; // waves per SIMD x instructions per wave
    printf("%-34s %.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
}

int main()
{
    float *d;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>(d, "v_mul_f32", 1);
    run<1>(d, "v_add_f32", 1);
    run<15>(d, "v_fma_f32", 1);
    run<2>(d, "v_pk_mul_f32", 1);
    run<3>(d, "v_pk_add_f32", 1);
    run<22>(d, "v_pk_fma_f32", 1);
    run<16>(d, "v_max3_f32", 1);
    run<14>(d, "v_trunc_f32", 1);
    run<9>(d, "v_cvt_f32_i32", 1);
    run<18>(d, "v_cvt_f32_ubyte1", 1);
    run<10>(d, "v_cvt_pk_u8_f32", 1);
    run<4>(d, "v_and_b32", 1);
    run<5>(d, "v_add_u32", 1);
    run<11>(d, "v_lshlrev_b32", 1);
    run<6>(d, "v_lshl_add_u32", 1);
    run<17>(d, "v_bfe_u32", 1);
    run<21>(d, "v_alignbit_b32", 1);
    run<7>(d, "v_ffbl_b32", 1);
    run<8>(d, "v_mov_b32", 1);
    run<12>(d, "v_cndmask_b32", 1);
    run<13>(d, "v_cmp_eq_u32", 1);
    run<19>(d, "v_mul_lo_u32", 1);
    run<20>(d, "v_mad_u32_u24", 1);
    run<23>(d, "v_mul_f32 + v_and_b32 (pair)", 2);
    return 0;
}
