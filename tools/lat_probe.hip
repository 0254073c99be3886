// Micro-benchmark (gfx950): what ONE wave pays per instruction — the numbers the entropy kernels' loops are designed on.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/lat_probe tools/lat_probe.hip && /tmp/lat_probe
// Every test is one workgroup of 64 lanes (one wave on the whole GPU, or NW waves on one CU for the LDS tests), a loop of
// N unrolled instructions timed with s_memtime; prints shader cycles per instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

__global__ void k_probe(unsigned long long *out, int mode, int iters)
{
    __shared__ unsigned lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = ((i * 2654435761u) >> 8) & 0x3FFC; // random byte offsets (4-aligned) < 16 KB
    __syncthreads();
    unsigned a = threadIdx.x * 4u, b = 3, c = 5, d = 7;
    unsigned long long t0 = 0, t1 = 0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < iters; it++) {
        if (mode == 0) { // dependent v_add_u32 chain
            REP64(asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(a) : "v"(b));)
        } else if (mode == 1) { // four independent chains, interleaved
            REP16(asm volatile("v_add_u32_e32 %0, %0, %4\n\tv_add_u32_e32 %1, %1, %4\n\tv_add_u32_e32 %2, %2, %4\n\tv_add_u32_e32 %3, %3, %4"
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(7u));)
        } else if (mode == 2) { // dependent chain through v_cndmask with vcc from a compare of the chain value
            REP16(asm volatile("v_cmp_lt_u32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_add_u32_e32 %0, 1, %0\n\tv_and_b32_e32 %0, 0xffff, %0" : "+v"(a) : "v"(b) : "vcc");)
        } else if (mode == 3) { // LDS pointer chase: ds_read_b32 -> address of the next
            REP16(asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a) : : "memory");)
        } else if (mode == 4) { // LDS pointer chase with a b128 read issued behind the b32
            REP16(asm volatile("ds_read_b32 %0, %0\n\tds_read_b128 v[120:123], %1\n\ts_waitcnt lgkmcnt(1)" : "+v"(a) : "v"(b * 16u) : "memory", "v120", "v121", "v122", "v123");)
        } else if (mode == 5) { // taken s_branch per instruction pair
            REP16(asm volatile("v_add_u32_e32 %0, %0, %1\n\ts_branch 1f\n\ts_nop 0\n1:" : "+v"(a) : "v"(b));)
        } else if (mode == 6) { // SALU dependent chain
            unsigned s = it;
            REP64(asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) : : "scc");)
            a += s;
        } else if (mode == 7) { // VALU compare -> SALU and -> VALU select (mask through an SGPR pair)
            unsigned long long m;
            REP16(asm volatile("v_cmp_lt_u32_e64 %2, %1, %0\n\ts_and_b64 %2, %2, exec\n\tv_cndmask_b32_e64 %0, %0, %1, %2\n\tv_add_u32_e32 %0, 1, %0" : "+v"(a), "+v"(b), "=&s"(m));)
        } else if (mode == 8) { // independent VALU, alternating with SALU
            unsigned s = it;
            REP16(asm volatile("v_add_u32_e32 %0, %0, %3\n\ts_add_u32 %2, %2, 3\n\tv_add_u32_e32 %1, %1, %3\n\ts_add_u32 %2, %2, 5" : "+v"(a), "+v"(b), "+s"(s) : "v"(7u) : "scc");)
            a += s;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a + b + c + d == 0x12345678u) out[1000] = a; // keep the chains alive
}

int main()
{
    unsigned long long *d, h[8];
    hipMalloc(&d, 8192 * sizeof(*d));
    const char *names[] = {"dependent v_add_u32", "4 independent v_add_u32 chains", "cmp -> cndmask -> add -> and (dependent)", "LDS chase ds_read_b32",
                           "LDS chase b32 + b128 behind it", "v_add + taken s_branch", "dependent s_add_u32", "v_cmp_e64 -> s_and -> v_cndmask_e64 -> v_add", "valu / salu alternating"};
    const int per[] = {64, 64, 64, 16, 16, 16, 64, 64, 64};
    for (int waves : {1, 4, 8, 16})
        for (int mode = 0; mode < 9; mode++) {
            const int iters = 200;
            k_probe<<<1, 64 * waves>>>(d, mode, iters);
            hipDeviceSynchronize();
            k_probe<<<1, 64 * waves>>>(d, mode, iters);
            hipDeviceSynchronize();
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("waves/CU %d  %-48s %7.2f cycles per %s\n", waves, names[mode], (double)h[0] / (iters * per[mode]),
                   mode == 3 || mode == 4 ? "read" : (mode == 5 ? "add+branch" : "instruction"));
        }
    return 0;
}
