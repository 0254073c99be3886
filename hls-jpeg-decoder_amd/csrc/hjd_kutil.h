// hjd_kutil.h — small device helpers shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hjd_device.h"

#define WAVE 64

// (Raised wave priority — s_setprio 3 — for the entropy kernels, whose waves issue an instruction every 8-12 cycles at best and
// queue for issue slots beside the back end's eight waves per SIMD: tried in round 4, the step got 1 % slower.)

// image whose [first, first+count) range contains idx; `field` selects the first-index member
template <typename F>
__device__ __forceinline__ int find_image(const HjdImage *img, int n, uint32_t idx, F first_of)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (first_of(img[mid]) <= idx) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & (WAVE - 1);
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        uint32_t t = __shfl_up(v, d, WAVE);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan over the block (blockDim.x multiple of 64, <= 1024); returns exclusive prefix,
// *total = block sum.  s_w: >= 17 words of LDS scratch.  Two barriers (every lane adds up the waves' totals itself; a serial
// pass by one thread between two more barriers was the first form: these scans sit in the prologues of latency-bound kernels).
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_w, uint32_t *total)
{
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == WAVE - 1) s_w[wv] = inc;
    __syncthreads();
    uint32_t pre = 0, tot = 0;
    for (int i = 0; i < nw; i++) {
        const uint32_t t = s_w[i];
        tot += t;
        pre += i < wv ? t : 0u;
    }
    *total = tot;
    __syncthreads();
    return inc - v + pre;
}

// two values at once: their sums over the block (one barrier; s2: >= 32 words that nothing else touches until the next barrier)
__device__ __forceinline__ void block_sum2(uint32_t &a, uint32_t &b, uint32_t *s2)
{
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int d = WAVE / 2; d >= 1; d >>= 1) {
        a += __shfl_xor(a, d, WAVE);
        b += __shfl_xor(b, d, WAVE);
    }
    if (lane == 0) s2[wv * 2] = a, s2[wv * 2 + 1] = b;
    __syncthreads();
    a = b = 0;
    for (int i = 0; i < nw; i++) a += s2[i * 2], b += s2[i * 2 + 1];
}

// ... and their exclusive prefixes over the block's lanes, in place; *ta, *tb = the block sums (one barrier; s2 as above)
__device__ __forceinline__ void block_excl_scan2(uint32_t &a, uint32_t &b, uint32_t *s2, uint32_t *ta, uint32_t *tb)
{
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t ia = a, ib = b;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t xa = __shfl_up(ia, d, WAVE), xb = __shfl_up(ib, d, WAVE);
        if (lane >= d) ia += xa, ib += xb;
    }
    if (lane == WAVE - 1) s2[wv * 2] = ia, s2[wv * 2 + 1] = ib;
    __syncthreads();
    uint32_t pa = 0, pb = 0, sa = 0, sb = 0;
    for (int i = 0; i < nw; i++) {
        const uint32_t xa = s2[i * 2], xb = s2[i * 2 + 1];
        sa += xa, sb += xb;
        pa += i < wv ? xa : 0u, pb += i < wv ? xb : 0u;
    }
    *ta = sa, *tb = sb;
    a = pa + ia - a;
    b = pb + ib - b;
}

