// Micro-benchmark: issue rate of v_pk_mul_f32 / v_pk_add_f32 against v_mul_f32 / v_add_f32 on gfx950, at full
// occupancy (8 waves per SIMD), independent accumulators.  Prints cycles per wave-instruction per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o /tmp/pk_rate tools/pk_rate.hip && /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float2v x[8];
    for (int i = 0; i < 8; i++) x[i] = (float2v){a + i + threadIdx.x, b + i};
    float2v m = {a, b};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) x[i] = x[i] * m;                       // v_pk_mul_f32
                else if (MODE == 1) x[i] = x[i] + m;                  // v_pk_add_f32
                else if (MODE == 2) { x[i].x = x[i].x * m.x; }        // v_mul_f32
                else { x[i].x = x[i].x * m.x; x[i].y = x[i].y * m.y; } // two v_mul_f32 (if not re-packed)
            }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += x[i].x + x[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float *d;
    hipMalloc(&d, 256 * 8 * 256 * 4 * 4);
    const int iters = 2000, grid = 256 * 8; // 8 blocks of 4 waves per CU = 8 waves per SIMD
    const char *names[4] = {"v_pk_mul_f32", "v_pk_add_f32", "v_mul_f32", "2 x v_mul_f32 (source level)"};
    for (int mode = 0; mode < 4; mode++) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.9999f);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.9999f);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.9999f);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.9999f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double insts_per_simd = 8.0 * iters * 32 * (mode == 3 ? 2 : 1); // waves per SIMD x instructions per wave
        printf("%-30s %.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", names[mode], ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
    }
    return 0;
}
