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
// *total = block sum.  s_w: >= 17 words of LDS scratch.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_w, uint32_t *total)
{
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == WAVE - 1) s_w[wv] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < nw; i++) {
            uint32_t t = s_w[i];
            s_w[i] = run;
            run += t;
        }
        s_w[16] = run;
    }
    __syncthreads();
    uint32_t r = inc - v + s_w[wv];
    *total = s_w[16];
    __syncthreads();
    return r;
}

