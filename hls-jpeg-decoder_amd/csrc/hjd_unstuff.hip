// hjd_unstuff.hip — byte-level preparation of the scan and the prefix sums between the two
// entropy passes (gfx950).
//   k_unstuff_count / k_unstuff_write  bit-reader byte rule FF00 -> FF      (loadjpg.cpp:468-482)
//   k_scan                             block index + DC predictor prefix     (loadjpg.cpp:619-667, :1159-1162)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hjd_device.h"
#include "hjd_kernels.h"
#include "hjd_kutil.h"

// ---- stage 0: unstuff ------------------------------------------------------------------------
// Bit-reader rule of FillNBits (loadjpg.cpp:468-482): the byte after a data 0xFF is dropped when
// it is 0x00.  Here that is done once, data-parallel: 16 raw bytes per lane, 4 KiB per workgroup.

__device__ __forceinline__ uint32_t unstuff_keepmask(const HjdImage &im, const uint8_t *raw, uint32_t lc,
                                                      uint4 *vout, uint8_t *s_last)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t off = lc * HJD_UNSTUFF_CHUNK + tid * 16;
    const uint8_t *base = raw + im.raw_off + off;
    uint4 v = *reinterpret_cast<const uint4 *>(base);
    *vout = v;
    s_last[tid] = (uint8_t)(v.w >> 24);
    __syncthreads();
    uint32_t prev;
    if (tid == 0) prev = (lc == 0) ? 0u : base[-1];
    else prev = s_last[tid - 1];
    int nvalid = (int)im.scan_len - (int)off;
    nvalid = nvalid < 0 ? 0 : (nvalid > 16 ? 16 : nvalid);
    uint32_t words[4] = {v.x, v.y, v.z, v.w};
    uint32_t mask = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        uint32_t b = (words[j >> 2] >> ((j & 3) * 8)) & 0xFF;
        bool keep = (j < nvalid) && !(b == 0 && prev == 0xFF);
        mask |= (keep ? 1u : 0u) << j;
        prev = b;
    }
    return mask;
}

__global__ __launch_bounds__(256) void k_unstuff_count(HjdParams P)
{
    __shared__ uint8_t s_last[256];
    __shared__ uint32_t s_w[17];
    const uint32_t c = blockIdx.x;
    const int ii = find_image(P.img, P.nimg, c, [](const HjdImage &m) { return m.chunk0; });
    const HjdImage &im = P.img[ii];
    uint4 v;
    uint32_t mask = unstuff_keepmask(im, P.raw, c - im.chunk0, &v, s_last);
    uint32_t total;
    block_excl_scan(__popc(mask), s_w, &total);
    if (threadIdx.x == 0) P.chunk_kept[c] = total;
}

__global__ __launch_bounds__(256) void k_unstuff_write(HjdParams P)
{
    __shared__ uint8_t s_last[256];
    __shared__ uint32_t s_w[17];
    const uint32_t c = blockIdx.x;
    const int ii = find_image(P.img, P.nimg, c, [](const HjdImage &m) { return m.chunk0; });
    const HjdImage &im = P.img[ii];
    const uint32_t lc = c - im.chunk0;
    // kept bytes of this image's earlier chunks
    uint32_t part = 0;
    for (uint32_t k = threadIdx.x; k < lc; k += blockDim.x) part += P.chunk_kept[im.chunk0 + k];
    uint32_t base;
    block_excl_scan(part, s_w, &base);
    uint4 v;
    uint32_t mask = unstuff_keepmask(im, P.raw, lc, &v, s_last);
    uint32_t total;
    uint32_t pos = base + block_excl_scan(__popc(mask), s_w, &total);
    // Transposed stream (see hjd_entropy.hip): word k of subsequence i -> ((i/64)*(wps+2) + k)*64 + i%64,
    // bytes MSB-first inside a word; words 0 and 1 of row i are repeated as words wps, wps+1 of row i-1.
    uint8_t *dst = P.ustream + im.ustream_off;
    const uint32_t lw = P.log2_words, wps = 1u << lw;
    uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (mask & (1u << j)) {
            const uint8_t bv = (uint8_t)(words[j >> 2] >> ((j & 3) * 8));
            const uint32_t wq = pos >> 2, i = wq >> lw, k = wq & (wps - 1u), bsel = 3u - (pos & 3u);
            dst[(((size_t)(i >> 6) * (wps + 2u) + k) * 64u + (i & 63u)) * 4u + bsel] = bv;
            if (k < 2u && i >= 1u) {
                const uint32_t i1 = i - 1u;
                dst[(((size_t)(i1 >> 6) * (wps + 2u) + wps + k) * 64u + (i1 & 63u)) * 4u + bsel] = bv;
            }
            pos++;
        }
    }
    if (lc == im.nchunks - 1 && threadIdx.x == 0) P.st[ii].ustream_len = base + total;
}

// ---- stage 2: prefix sums over subsequences ---------------------------------------------------
// block index and DC predictors at every subsequence start; DC prediction runs over the whole
// image and is never reset (loadjpg.cpp:1159-1162), all arithmetic mod 2^16 (loadjpg.cpp:664-665)

__global__ __launch_bounds__(1024) void k_scan(HjdParams P)
{
    __shared__ uint32_t s_w[17];
    const int ii = blockIdx.x;
    const HjdImage &im = P.img[ii];
    const uint32_t S = P.subseq_bits;
    const uint32_t total_bits = P.st[ii].ustream_len * 8u;
    const uint32_t nsub = (total_bits + S - 1) / S;
    const uint32_t per = (nsub + blockDim.x - 1) / blockDim.x;
    const uint32_t lo = min(threadIdx.x * per, nsub), hi = min(lo + per, nsub);
    uint32_t a = 0, b = 0, c = 0, d = 0;
    for (uint32_t k = lo; k < hi; k++) {
        const HjdSub r = P.sub[im.sub0 + k];
        a += r.nblk;
        b += (uint32_t)(int)r.dc[0];
        c += (uint32_t)(int)r.dc[1];
        d += (uint32_t)(int)r.dc[2];
    }
    uint32_t ta, tb, tc, td;
    a = block_excl_scan(a, s_w, &ta);
    b = block_excl_scan(b, s_w, &tb);
    c = block_excl_scan(c, s_w, &tc);
    d = block_excl_scan(d, s_w, &td);
    for (uint32_t k = lo; k < hi; k++) {
        const HjdSub r = P.sub[im.sub0 + k];
        HjdSubStart s;
        s.blk0 = a;
        s.pred[0] = (int16_t)b;
        s.pred[1] = (int16_t)c;
        s.pred[2] = (int16_t)d;
        s.pad = 0;
        P.substart[im.sub0 + k] = s;
        a += r.nblk;
        b += (uint32_t)(int)r.dc[0];
        c += (uint32_t)(int)r.dc[1];
        d += (uint32_t)(int)r.dc[2];
    }
    if (threadIdx.x == 0) {
        P.st[ii].total_blocks = ta;
        if (ta < im.nblocks) atomicOr(&P.st[ii].flags, HJD_FLAG_SHORT);
    }
}

#define HJD_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e_ = hipGetLastError();         \
        if (e_ != hipSuccess) return (int)e_;      \
    } while (0)

int hjd_launch_unstuff(const HjdParams &P, uint32_t nchunks, hipStream_t s)
{
    if (!nchunks) return 0;
    hipLaunchKernelGGL(k_unstuff_count, dim3(nchunks), dim3(256), 0, s, P);
    HJD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_unstuff_write, dim3(nchunks), dim3(256), 0, s, P);
    HJD_LAUNCH_CHECK();
    return 0;
}

int hjd_launch_scan(const HjdParams &P, hipStream_t s)
{
    if (!P.nimg) return 0;
    hipLaunchKernelGGL(k_scan, dim3(P.nimg), dim3(1024), 0, s, P);
    HJD_LAUNCH_CHECK();
    return 0;
}
