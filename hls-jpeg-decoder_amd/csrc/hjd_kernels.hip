// hjd_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the baseline-JPEG decode path.
//
// Hot path being replaced: everything reachable from JpegDecodeHW (reference
// src/loadjpg.cpp:1134-1190).  Stage map:
//   k_unstuff_count / k_unstuff_write  bit-reader byte rule FF00 -> FF      (loadjpg.cpp:468-482)
//   k_sync                             ProcessHuffmanBlock, speculative      (loadjpg.cpp:499-863)
//   k_scan                             block index + DC predictor prefix     (loadjpg.cpp:619-667, :1159-1162)
//   k_write                            ProcessHuffmanBlock, final pass -> coefficient tiles
//   k_idct_color                       DecodeSingleBlock + YCrCB_to_RGB24_Block8x8 (+ WriteBMP24 layout)
//                                      (loadjpg.cpp:184-228, :867-932; openjpg.cpp:504-570)
//
// The scan of a baseline JPEG is one bit-serial string.  It is cut into fixed-size
// SUBSEQUENCES (S bits), one per lane.  Huffman streams self-synchronise: a decoder started at
// a wrong position/state soon falls into step with the true decode.  Each lane decodes its
// subsequence from a guessed start state, then repeatedly from the end state of its
// predecessor, until start == predecessor's end everywhere; by induction from subsequence 0
// (whose start is known) every state is then the true one.  Outputs are written only in the
// final pass, so speculation never reaches them.
//
// Compile with -ffp-contract=off: the reference's IDCT and colour arithmetic are fp32 with
// one rounding per operation; an FMA would change output bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hjd_device.h"
#include "hjd_kernels.h"

#define WAVE 64

// ---- small helpers ---------------------------------------------------------------------------

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
    uint8_t *dst = P.ustream + im.ustream_off;
    uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (mask & (1u << j)) {
            dst[pos ^ 3u] = (uint8_t)(words[j >> 2] >> ((j & 3) * 8)); // MSB-first words
            pos++;
        }
    }
    if (lc == im.nchunks - 1 && threadIdx.x == 0) P.st[ii].ustream_len = base + total;
}

// ---- entropy decode core ---------------------------------------------------------------------

struct SlowTab { // canonical tables for codes longer than HJD_LUT_BITS, in LDS
    uint16_t mincode[HJD_SLOTS][17];
    uint16_t count[HJD_SLOTS][17];
    uint16_t first[HJD_SLOTS][17];
    uint16_t sym_entry[HJD_SLOTS][256];
};

__device__ __forceinline__ void load_tables(const HjdTableSet *ts, uint16_t *s_lut, SlowTab *s_slow)
{
    // lut: 6*1024 u16 = 12 KiB = 768 uint4; slow: copied as u16 words
    const uint4 *src = reinterpret_cast<const uint4 *>(&ts->lut[0][0]);
    uint4 *dst = reinterpret_cast<uint4 *>(s_lut);
    for (uint32_t k = threadIdx.x; k < HJD_SLOTS * HJD_LUT_SIZE * 2 / 16; k += blockDim.x) dst[k] = src[k];
    const uint16_t *ss = &ts->mincode[0][0];
    uint16_t *sd = reinterpret_cast<uint16_t *>(s_slow);
    const uint32_t nslow = sizeof(SlowTab) / 2;
    for (uint32_t k = threadIdx.x; k < nslow; k += blockDim.x) sd[k] = ss[k];
}

// The workgroup's span of the bit string (HJD_SYNC_WG subsequences + a few words of overshoot)
// is staged once into LDS with coalesced 16-byte loads; lanes then read their words from LDS.
// Row r (one subsequence = 2^lw words) is padded by one word so that lanes reading the same
// column of their rows hit different banks.
struct StreamLds {
    const uint32_t *s; // staged words
    uint32_t w0;       // image-relative index of the first staged word
    uint32_t lw;       // log2(words per subsequence)
    __device__ __forceinline__ uint32_t word(uint32_t widx) const
    {
        const uint32_t r = widx - w0;
        return s[r + (r >> lw)];
    }
};

__device__ __forceinline__ void stage_stream(const uint32_t *__restrict__ w, uint32_t w0, uint32_t lw, uint32_t *s_stream)
{
    const uint32_t nwords = ((uint32_t)HJD_SYNC_WG << lw) + 8; // span + overshoot (31 bits + 2 prefetched words)
    const uint4 *src = reinterpret_cast<const uint4 *>(w + w0);
    for (uint32_t k = threadIdx.x; k < nwords / 4; k += blockDim.x) {
        const uint4 v = src[k];
        const uint32_t i0 = 4 * k;
        s_stream[i0 + (i0 >> lw)] = v.x;
        s_stream[i0 + 1 + ((i0 + 1) >> lw)] = v.y;
        s_stream[i0 + 2 + ((i0 + 2) >> lw)] = v.z;
        s_stream[i0 + 3 + ((i0 + 3) >> lw)] = v.w;
    }
}

// Decode symbols from state (p,z,m) until p >= end.  One symbol per iteration; the only
// divergent paths are the long-code search and (WRITE) the stores.
//   WRITE = false: nblk counts blocks completed, dc0..2 sum the DC differences   (synchronisation)
//   WRITE = true : nblk is the absolute block index, dc0..2 the running predictors; coefficients
//                  are stored in zig-zag order (the DCT[] array of loadjpg.cpp:499-863)
template <bool WRITE>
__device__ __forceinline__ void decode_span(const StreamLds &sl, const uint16_t *s_lut, const SlowTab *s_slow,
                                            uint32_t nby, uint32_t nb, uint32_t &p, uint32_t &z, uint32_t &m,
                                            uint32_t end, uint32_t &nblk, int &dc0, int &dc1, int &dc2,
                                            int16_t *__restrict__ coef, uint32_t nblocks, uint32_t &flags)
{
    uint32_t wi = p >> 5, o = p & 31;
    uint32_t hi = sl.word(wi), lo = sl.word(wi + 1);
    wi += 2;
    uint32_t comp = m < nby ? 0u : m - nby + 1u;
    while (p < end && (!WRITE || nblk < nblocks)) {
        const uint32_t nxt = sl.word(wi); // next word, fetched alongside the table lookup
        const uint32_t win = __funnelshift_l(lo, hi, o);
        const uint32_t slot = comp * 2u + (z != 0u ? 1u : 0u);
        const uint32_t e = s_lut[slot * HJD_LUT_SIZE + (win >> (32 - HJD_LUT_BITS))];
        uint32_t len = e & 31u, s, adv;
        if (len == 0u) {
            // code longer than the primary table: canonical search (IsInHuffmanCodes, loadjpg.cpp:335-392,
            // restricted to the lengths the table cannot hold)
            const uint32_t v16 = win >> 16;
            len = 16u;
            s = 0u;
            adv = 0u;
            bool found = false;
            for (uint32_t k = HJD_LUT_BITS + 1; k <= 16u; k++) {
                const uint32_t d = (v16 >> (16u - k)) - s_slow->mincode[slot][k];
                if (d < s_slow->count[slot][k]) {
                    const uint32_t se = s_slow->sym_entry[slot][s_slow->first[slot][k] + d];
                    s = se & 15u;
                    adv = se >> 4;
                    len = k;
                    found = true;
                    break;
                }
            }
            if (WRITE && !found) flags |= HJD_FLAG_BAD_CODE;
        } else {
            s = (e >> 5) & 15u;
            adv = e >> 9;
        }
        // value bits + DetermineSign (loadjpg.cpp:396-409)
        const uint32_t vb = ((win << len) >> 1) >> (31u - s);
        const int val = (int)vb - ((vb < ((1u << s) >> 1)) ? (int)((1u << s) - 1u) : 0);
        const bool isdc = z == 0u;
        const int dval = isdc ? val : 0; // DC: loadjpg.cpp:619-667
        dc0 += comp == 0u ? dval : 0;
        dc1 += comp == 1u ? dval : 0;
        dc2 += comp == 2u ? dval : 0;
        if (WRITE) {
            const uint32_t zw = isdc ? 0u : z + adv - 1u; // AC run/size: loadjpg.cpp:776-808
            const int16_t cv = isdc ? (int16_t)(comp == 0u ? dc0 : (comp == 1u ? dc1 : dc2)) : (int16_t)val;
            if (isdc || s != 0u) {
                if (zw <= 63u) coef[(size_t)nblk * 64 + zw] = cv;
                else flags |= HJD_FLAG_OVERRUN;
            }
        }
        z += adv;
        if (z >= 64u) { // EOB or last coefficient: next block (DecodeMCU order, loadjpg.cpp:945-997)
            z = 0u;
            nblk++;
            m++;
            if (m == nb) m = 0u;
            comp = m < nby ? 0u : m - nby + 1u;
        }
        const uint32_t tot = len + s;
        p += tot;
        o += tot;
        const bool roll = o >= 32u;
        hi = roll ? lo : hi;
        lo = roll ? nxt : lo;
        wi += roll ? 1u : 0u;
        o &= 31u;
    }
}

// ---- stage 1: synchronisation ----------------------------------------------------------------

__global__ __launch_bounds__(HJD_SYNC_WG) void k_sync(HjdParams P, int launch_idx)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_lut[HJD_SLOTS * HJD_LUT_SIZE];
    __shared__ __attribute__((aligned(16))) SlowTab s_slow;
    __shared__ uint32_t s_ep[HJD_SYNC_WG], s_ezm[HJD_SYNC_WG];
    extern __shared__ __attribute__((aligned(16))) uint32_t s_stream[];

    const uint32_t tid = threadIdx.x;
    const int ii = find_image(P.img, P.nimg, blockIdx.x, [](const HjdImage &m) { return m.syncwg0; });
    const HjdImage &im = P.img[ii];
    load_tables(&P.tabs[im.tabset], s_lut, &s_slow);

    const uint32_t S = P.subseq_bits;
    const uint32_t lw = P.log2_words;
    const StreamLds sl = {s_stream, ((blockIdx.x - im.syncwg0) * HJD_SYNC_WG) << lw, lw};
    stage_stream(reinterpret_cast<const uint32_t *>(P.ustream + im.ustream_off), sl.w0, lw, s_stream);
    const uint32_t total_bits = P.st[ii].ustream_len * 8u;
    const uint32_t nsub = (total_bits + S - 1) / S;
    const uint32_t i = (blockIdx.x - im.syncwg0) * HJD_SYNC_WG + tid;
    const bool valid = i < nsub;
    const uint32_t g = im.sub0 + i;
    const uint32_t sp = i * S;
    const uint32_t end = min(sp + S, total_bits);
    const uint32_t nby = im.nby, nb = im.nb;

    uint32_t up = sp, uzm = 0, ep = sp, ezm = 0, nblk = 0, flags = 0;
    int dc0 = 0, dc1 = 0, dc2 = 0;
    uint32_t ext_p = sp, ext_zm = 0; // start state offered to lane 0 from outside the workgroup
    __syncthreads();

    if (valid) {
        if (launch_idx == 0) {
            uint32_t p = sp, z = 0, m = 0;
            decode_span<false>(sl, s_lut, &s_slow, nby, nb, p, z, m, end, nblk, dc0, dc1, dc2, nullptr, 0, flags);
            ep = p;
            ezm = z | (m << 8);
        } else {
            const HjdSub r = P.sub[g];
            ep = r.e_p;
            ezm = r.e_zm;
            up = r.u_p;
            uzm = r.u_zm;
            nblk = r.nblk;
            dc0 = r.dc[0];
            dc1 = r.dc[1];
            dc2 = r.dc[2];
            if (tid == 0 && i > 0) {
                const uint2 pe = *reinterpret_cast<const uint2 *>(&P.sub[g - 1]);
                ext_p = pe.x;
                ext_zm = pe.y;
            }
        }
    }
    s_ep[tid] = ep;
    s_ezm[tid] = ezm;
    __syncthreads();

    uint32_t ndec = (valid && launch_idx == 0) ? 1u : 0u;
    int iters = 0;
    for (int iter = 0; iter <= HJD_SYNC_WG; iter++) {
        iters = iter;
        uint32_t wp, wzm;
        if (i == 0) {
            wp = 0;
            wzm = 0; // the true start of the scan (loadjpg.cpp:1148-1149)
        } else if (tid == 0) {
            wp = ext_p;
            wzm = ext_zm;
        } else {
            wp = s_ep[tid - 1];
            wzm = s_ezm[tid - 1];
        }
        const bool changed = valid && (wp != up || wzm != uzm);
        if (changed) {
            uint32_t p = wp, z = wzm & 0xFF, m = wzm >> 8;
            nblk = 0;
            dc0 = dc1 = dc2 = 0;
            decode_span<false>(sl, s_lut, &s_slow, nby, nb, p, z, m, end, nblk, dc0, dc1, dc2, nullptr, 0, flags);
            ep = p;
            ezm = z | (m << 8);
            up = wp;
            uzm = wzm;
            ndec++;
        }
        __syncthreads();
        if (changed) {
            s_ep[tid] = ep;
            s_ezm[tid] = ezm;
        }
        if (!__syncthreads_or(changed ? 1 : 0)) break;
    }

    if (valid) {
        HjdSub r;
        r.e_p = ep;
        r.e_zm = ezm;
        r.u_p = up;
        r.u_zm = uzm;
        r.nblk = nblk;
        r.dc[0] = (int16_t)dc0;
        r.dc[1] = (int16_t)dc1;
        r.dc[2] = (int16_t)dc2;
        r.pad = 0;
        P.sub[g] = r;
    }
    // diagnostics
    atomicAdd(&P.st[ii].sync_decodes, ndec);
    if (tid == 0) {
        atomicAdd(&P.st[ii].sync_iters, (uint32_t)iters);
        atomicMax(&P.st[ii].sync_iters_max, (uint32_t)iters);
    }
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

// ---- stage 3: final entropy decode, coefficients out ------------------------------------------

__global__ __launch_bounds__(HJD_SYNC_WG) void k_write(HjdParams P)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_lut[HJD_SLOTS * HJD_LUT_SIZE];
    __shared__ __attribute__((aligned(16))) SlowTab s_slow;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_stream[];

    const uint32_t tid = threadIdx.x;
    const int ii = find_image(P.img, P.nimg, blockIdx.x, [](const HjdImage &m) { return m.syncwg0; });
    const HjdImage &im = P.img[ii];
    load_tables(&P.tabs[im.tabset], s_lut, &s_slow);
    const uint32_t lw = P.log2_words;
    const StreamLds sl = {s_stream, ((blockIdx.x - im.syncwg0) * HJD_SYNC_WG) << lw, lw};
    stage_stream(reinterpret_cast<const uint32_t *>(P.ustream + im.ustream_off), sl.w0, lw, s_stream);
    __syncthreads();

    const uint32_t S = P.subseq_bits;
    const uint32_t total_bits = P.st[ii].ustream_len * 8u;
    const uint32_t nsub = (total_bits + S - 1) / S;
    const uint32_t i = (blockIdx.x - im.syncwg0) * HJD_SYNC_WG + tid;
    if (i >= nsub) return;
    const uint32_t g = im.sub0 + i;
    const uint32_t end = min(i * S + S, total_bits);

    uint32_t p = 0, zm = 0;
    if (i > 0) {
        const uint2 pe = *reinterpret_cast<const uint2 *>(&P.sub[g - 1]);
        p = pe.x;
        zm = pe.y;
    }
    uint32_t flags = 0;
    const HjdSub me = P.sub[g];
    if (me.u_p != p || me.u_zm != zm) flags |= HJD_FLAG_NOT_CONVERGED; // induction step failed: host re-synchronises
    const HjdSubStart st = P.substart[g];
    uint32_t z = zm & 0xFF, m = zm >> 8, blk = st.blk0;
    int dc0 = st.pred[0], dc1 = st.pred[1], dc2 = st.pred[2];
    decode_span<true>(sl, s_lut, &s_slow, im.nby, im.nb, p, z, m, end, blk, dc0, dc1, dc2, P.coef + im.coef_off,
                      im.nblocks, flags);
    if (flags) atomicOr(&P.st[ii].flags, flags);
}

// ---- stage 4: dequantise + IDCT + upsample + colour + store -----------------------------------

// natural index -> zig-zag index (loadjpg.cpp:56-66)
__constant__ uint8_t c_zz[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42,
                                 3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
                                 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
                                 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

// COS[x][u] = cosf(fl((2x+1)*u*3.14f)/16) exactly as the reference's x86-64/glibc-2.35 build
// evaluates it (loadjpg.cpp:108,120).  A table, never recomputed: entry [6][7]
// is libm-specific, and PI = 3.14f makes the basis non-orthogonal, so no fast IDCT is bit-exact.
__constant__ float c_cos[64] = {
    0x1p+0f, 0x1.f62c08p-1f, 0x1.d910b8p-1f, 0x1.a9cc2p-1f, 0x1.6a2ecap-1f, 0x1.1ca9eep-1f, 0x1.886ecap-2f, 0x1.90f1bep-3f,
    0x1p+0f, 0x1.a9cc2p-1f, 0x1.886ecap-2f, -0x1.8dbedp-3f, -0x1.699b2p-1f, -0x1.f6032cp-1f, -0x1.d9606ap-1f, -0x1.1d5756p-1f,
    0x1p+0f, 0x1.1ca9eep-1f, -0x1.86ed1p-2f, -0x1.f6032cp-1f, -0x1.6ac236p-1f, 0x1.8a8b92p-3f, 0x1.d8706ep-1f, 0x1.aab36ap-1f,
    0x1p+0f, 0x1.90f1bep-3f, -0x1.d8c0bap-1f, -0x1.1d5756p-1f, 0x1.690742p-1f, 0x1.aab36ap-1f, -0x1.83e8d8p-2f, -0x1.f6a4aep-1f,
    0x1p+0f, -0x1.8dbedp-3f, -0x1.d9606ap-1f, 0x1.1b4e94p-1f, 0x1.6b556cp-1f, -0x1.a86f2p-1f, -0x1.8cf256p-2f, 0x1.f5869ep-1f,
    0x1p+0f, -0x1.1bfc56p-1f, -0x1.89f044p-2f, 0x1.f67ccap-1f, -0x1.68731cp-1f, -0x1.9a88c4p-3f, 0x1.da4da2p-1f, -0x1.a7851ep-1f,
    0x1p+0f, -0x1.a9581p-1f, 0x1.856b06p-2f, 0x1.9756cp-3f, -0x1.6be866p-1f, 0x1.f6cc3cp-1f, -0x1.d77dbp-1f, 0x1.1895aep-1f,
    0x1p+0f, -0x1.f6032cp-1f, 0x1.d8706ep-1f, -0x1.a86f2p-1f, 0x1.67dec4p-1f, -0x1.19442cp-1f, 0x1.7f609p-2f, -0x1.7a87f6p-3f};

#define MAXGB (HJD_GROUP_MCUS * 6) // blocks per workgroup

__device__ __forceinline__ uint32_t clamp255(int v) { return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// ConvertYCrCbtoRGB as called at loadjpg.cpp:918 (:867-880), one rounding per operation
__device__ __forceinline__ void ycc_to_rgb(int y, int cb, int cr, uint32_t &r, uint32_t &g, uint32_t &b)
{
    const float fy = (float)y, fcb = (float)(cb - 128), fcr = (float)(cr - 128);
    const float red = fy + 1.402f * fcr;
    const float green = (fy - 0.34414f * fcb) - 0.71414f * fcr;
    const float blue = fy + 1.772f * fcb;
    r = clamp255((int)red);
    g = clamp255((int)green);
    b = clamp255((int)blue);
}

__global__ __launch_bounds__(256) void k_idct_color(HjdParams P)
{
    __shared__ int16_t s_coef[MAXGB][64]; // dequantised, index u*8+v (u = horizontal frequency)
    __shared__ uint64_t s_mask[MAXGB];    // bit u*8+v set where the coefficient is non-zero
    __shared__ __attribute__((aligned(4))) uint8_t s_pix[MAXGB][64]; // decoded 8x8 tiles, row-major
    __shared__ float s_cosT[64];          // [u][x] = COS[x][u]
    __shared__ float s_cos[64];           // [v][y] = COS[y][v]  (same matrix, named for its use)
    __shared__ uint8_t s_qt[3][64];
    __shared__ uint8_t s_pos[64];

    const uint32_t tid = threadIdx.x;
    const int ii = find_image(P.img, P.nimg, blockIdx.x, [](const HjdImage &m) { return m.group0; });
    const HjdImage &im = P.img[ii];
    const uint32_t lg = blockIdx.x - im.group0;
    const uint32_t mrow = lg / im.groups_per_row, gx = lg % im.groups_per_row;
    const uint32_t mcu_first = gx * HJD_GROUP_MCUS;
    const uint32_t K = min((uint32_t)HJD_GROUP_MCUS, im.mcux - mcu_first);
    const uint32_t nb = im.nb, nby = im.nby, hf = im.hf, vf = im.vf;
    const uint32_t KB = K * nb;
    const uint32_t b0 = (mrow * im.mcux + mcu_first) * nb;

    if (tid < 64) {
        s_cosT[tid] = c_cos[(tid & 7) * 8 + (tid >> 3)];
        s_cos[tid] = c_cos[(tid & 7) * 8 + (tid >> 3)];
    }
    if (tid < 192) s_qt[tid >> 6][tid & 63] = im.qt[tid >> 6][tid & 63];
    if (tid < 64) s_pos[c_zz[tid]] = (uint8_t)((tid & 7) * 8 + (tid >> 3)); // zig-zag slot -> u*8+v (natural tid = v*8+u)
    if (tid < MAXGB) s_mask[tid] = 0;
    __syncthreads();

    // phase 1: DequantizeBlock + DeZigZag + TransformArray (loadjpg.cpp:144-180).  The group's
    // coefficient rows are contiguous: one coalesced 16-byte load (8 zig-zag slots) per thread and
    // step; only non-zero values are scattered to their (u,v) slot and flagged in the block's mask.
    // int16 wrap of the product as the reference's (short)(int) cast.
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(P.coef + im.coef_off + (size_t)b0 * 64);
        const uint32_t n4 = KB * 8;
        for (uint32_t k = tid; k < n4; k += 256) {
            const uint4 v = src[k];
            if ((v.x | v.y | v.z | v.w) == 0u) continue;
            const uint32_t blk = k >> 3, z0 = (k & 7) * 8;
            const uint32_t mm = blk % nb;
            const uint32_t comp = mm < nby ? 0u : mm - nby + 1u;
            const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
            uint64_t mk = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int c = (int)(int16_t)(wd[j >> 1] >> ((j & 1) * 16));
                if (c != 0) {
                    const uint32_t zz = z0 + j;
                    const int16_t nat = (int16_t)(c * (int)s_qt[comp][zz]);
                    if (nat != 0) {
                        const uint32_t pos = s_pos[zz];
                        s_coef[blk][pos] = nat;
                        mk |= 1ull << pos;
                    }
                }
            }
            if (mk) atomicOr(reinterpret_cast<unsigned long long *>(&s_mask[blk]), (unsigned long long)mk);
        }
    }
    __syncthreads();

    // phase 2: IDCT_calc / PerformIDCT (loadjpg.cpp:105-140).  Thread = (block, output column x),
    // 8 row accumulators.  Terms are added in the reference's order (u outer, v inner); zero
    // coefficients add +-0 and are skipped.  term = (((C(u)C(v)) * b) * COS[x][u]) * COS[y][v].
    {
        const uint32_t x = tid & 7;
        for (uint32_t blk = tid >> 3; blk < KB; blk += 32) {
            float acc[8];
#pragma unroll
            for (int y = 0; y < 8; y++) acc[y] = 0.0f;
            uint64_t mk = s_mask[blk];
            while (mk) {
                const uint32_t j = (uint32_t)__ffsll((unsigned long long)mk) - 1u;
                mk &= mk - 1;
                const uint32_t u = j >> 3, v = j & 7;
                // C(u)*C(v): fl(C0*C0) = 0x1.fffffep-2 (not 0.5), C0 = 1/sqrtf(2) = 0x1.6a09e6p-1
                const float kk = (u == 0) ? (v == 0 ? 0x1.fffffep-2f : 0x1.6a09e6p-1f) : (v == 0 ? 0x1.6a09e6p-1f : 1.0f);
                const float t1 = kk * (float)s_coef[blk][j];
                const float t2 = t1 * s_cosT[u * 8 + x];
                const float4 ca = *reinterpret_cast<const float4 *>(&s_cos[v * 8]);
                const float4 cb = *reinterpret_cast<const float4 *>(&s_cos[v * 8 + 4]);
                acc[0] = acc[0] + t2 * ca.x;
                acc[1] = acc[1] + t2 * ca.y;
                acc[2] = acc[2] + t2 * ca.z;
                acc[3] = acc[3] + t2 * ca.w;
                acc[4] = acc[4] + t2 * cb.x;
                acc[5] = acc[5] + t2 * cb.y;
                acc[6] = acc[6] + t2 * cb.z;
                acc[7] = acc[7] + t2 * cb.w;
            }
#pragma unroll
            for (int y = 0; y < 8; y++) {
                // (int)(0.25*sum) then short +128 then Clamp (loadjpg.cpp:123,136-137,219)
                const int16_t t = (int16_t)(int)(acc[y] * 0.25f);
                const int16_t lv = (int16_t)(t + 128);
                s_pix[blk][y * 8 + x] = (uint8_t)clamp255(lv);
            }
        }
    }
    __syncthreads();

    // phase 3: YCrCB_to_RGB24_Block8x8 (loadjpg.cpp:884-932): nearest-neighbour chroma, clip at
    // the image edge, then either m_rgb layout or WriteBMP24 layout (openjpg.cpp:555-568)
    {
        const uint32_t W = im.width, H = im.height;
        const uint32_t mw = 8 * hf, mh = 8 * vf;
        const uint32_t upr = K * 2 * hf; // 4-pixel units per row
        const uint32_t nunits = upr * mh;
        uint8_t *out = P.out + im.out_off;
        const bool bmp = im.out_format == 1;
        const uint32_t stride = im.out_stride;
        for (uint32_t t = tid; t < nunits; t += 256) {
            const uint32_t row = t / upr, ux = t % upr;
            const uint32_t px = ux * 4;
            const uint32_t k = px / mw, xin = px % mw;
            const uint32_t gy = mrow * mh + row;
            const uint32_t gxp = (mcu_first + k) * mw + xin;
            if (gy >= H || gxp >= W) continue; // loadjpg.cpp:907-908
            const uint32_t yblk = k * nb + (row >> 3) * hf + (xin >> 3);
            const uint32_t ypix = *reinterpret_cast<const uint32_t *>(&s_pix[yblk][(row & 7) * 8 + (xin & 7)]);
            const uint32_t cy = row / vf;
            const uint8_t *pcb = &s_pix[k * nb + nby][cy * 8];
            const uint8_t *pcr = &s_pix[k * nb + nby + 1][cy * 8];
            uint32_t rr[4], gg[4], bb[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t cx = (xin + j) / hf; // loadjpg.cpp:912
                ycc_to_rgb((int)((ypix >> (8 * j)) & 0xFF), (int)pcb[cx], (int)pcr[cx], rr[j], gg[j], bb[j]);
            }
            const uint32_t nvalid = min(4u, W - gxp);
            uint8_t *dst;
            if (bmp) dst = out + 54 + (size_t)(H - 1 - gy) * stride + (size_t)gxp * 3;
            else dst = out + (size_t)gy * stride + (size_t)gxp * 3;
            uint32_t c0[4], c2[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                c0[j] = bmp ? bb[j] : rr[j];
                c2[j] = bmp ? rr[j] : bb[j];
            }
            if (nvalid == 4 && ((reinterpret_cast<uintptr_t>(dst) & 3) == 0)) {
                uint32_t *d4 = reinterpret_cast<uint32_t *>(dst);
                d4[0] = c0[0] | (gg[0] << 8) | (c2[0] << 16) | (c0[1] << 24);
                d4[1] = gg[1] | (c2[1] << 8) | (c0[2] << 16) | (gg[2] << 24);
                d4[2] = c2[2] | (c0[3] << 8) | (gg[3] << 16) | (c2[3] << 24);
            } else {
                for (uint32_t j = 0; j < nvalid; j++) {
                    dst[j * 3 + 0] = (uint8_t)c0[j];
                    dst[j * 3 + 1] = (uint8_t)gg[j];
                    dst[j * 3 + 2] = (uint8_t)c2[j];
                }
            }
            if (bmp && gxp + nvalid == W) { // row padding (openjpg.cpp:563-567), written as zeros
                for (uint32_t q = 3 * W; q < stride; q++) out[54 + (size_t)(H - 1 - gy) * stride + q] = 0;
            }
        }
        if (bmp && lg == 0 && tid < 54) {
            // stBMFH (openjpg.cpp:507-547): 'BM', file size, 0, 0, 54, 40, W, H, planes 1, 24 bpp, rest 0
            const uint32_t fsz = W * H * 3 + H * (stride - 3 * W) + 54;
            uint32_t v = 0;
            if (tid == 0) v = 'B';
            else if (tid == 1) v = 'M';
            else if (tid >= 2 && tid < 6) v = (fsz >> (8 * (tid - 2))) & 0xFF;
            else if (tid == 10) v = 54;
            else if (tid == 14) v = 40;
            else if (tid >= 18 && tid < 22) v = (W >> (8 * (tid - 18))) & 0xFF;
            else if (tid >= 22 && tid < 26) v = (H >> (8 * (tid - 22))) & 0xFF;
            else if (tid == 26) v = 1;
            else if (tid == 28) v = 24;
            out[tid] = (uint8_t)v;
        }
    }
}

// ---- launch wrappers --------------------------------------------------------------------------

#define HJD_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t e_ = hipGetLastError();         \
        if (e_ != hipSuccess) return (int)e_;      \
    } while (0)

static inline uint32_t hjd_stream_lds_bytes(uint32_t lw)
{
    const uint32_t nwords = ((uint32_t)HJD_SYNC_WG << lw) + 8;
    return (nwords + (nwords >> lw) + 4) * 4;
}

int hjd_launch_unstuff(const HjdParams &P, uint32_t nchunks, hipStream_t s)
{
    if (!nchunks) return 0;
    hipLaunchKernelGGL(k_unstuff_count, dim3(nchunks), dim3(256), 0, s, P);
    HJD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_unstuff_write, dim3(nchunks), dim3(256), 0, s, P);
    HJD_LAUNCH_CHECK();
    return 0;
}

int hjd_launch_sync(const HjdParams &P, uint32_t nwg, int launch_idx, hipStream_t s)
{
    if (!nwg) return 0;
    const uint32_t lds = hjd_stream_lds_bytes(P.log2_words);
    if (lds > 40000 && hipFuncSetAttribute(reinterpret_cast<const void *>(k_sync), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_sync, dim3(nwg), dim3(HJD_SYNC_WG), lds, s, P, launch_idx);
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

int hjd_launch_write(const HjdParams &P, uint32_t nwg, hipStream_t s)
{
    if (!nwg) return 0;
    const uint32_t lds = hjd_stream_lds_bytes(P.log2_words);
    if (lds > 40000 && hipFuncSetAttribute(reinterpret_cast<const void *>(k_write), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_write, dim3(nwg), dim3(HJD_SYNC_WG), lds, s, P);
    HJD_LAUNCH_CHECK();
    return 0;
}

int hjd_launch_idct_color(const HjdParams &P, uint32_t ngroups, hipStream_t s)
{
    if (!ngroups) return 0;
    hipLaunchKernelGGL(k_idct_color, dim3(ngroups), dim3(256), 0, s, P);
    HJD_LAUNCH_CHECK();
    return 0;
}
