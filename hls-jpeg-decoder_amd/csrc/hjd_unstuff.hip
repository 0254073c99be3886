// hjd_unstuff.hip — byte-level preparation of the scan and the prefix sums between the two
// entropy passes (gfx950).
//   k_unstuff_count / k_unstuff_write  bit-reader byte rule FF00 -> FF      (loadjpg.cpp:468-482)
//   k_dcsum / k_dcapply                DC prediction as a prefix sum          (loadjpg.cpp:619-667, :1159-1162)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hjd_device.h"
#include "hjd_kernels.h"
#include "hjd_kutil.h"

// ---- stage 0: unstuff ------------------------------------------------------------------------
// Bit-reader rule of FillNBits (loadjpg.cpp:468-482): the byte after a data 0xFF is dropped when
// it is 0x00.  Here that is done once, data-parallel: 16 raw bytes per lane, 4 KiB per workgroup.
// Without DRI, "FF xx" (xx != 0) is ordinary data for the reference's reader and stays in the stream.
// With DRI the reference resynchronises on any FF xx it finds at a block start (loadjpg.cpp:535-550):
// such pairs are taken OUT of the stream and their positions recorded (mark mask, bit j = the pair's
// second byte), so that the entropy kernels can treat the intervals between them separately.
//
// Only a lane that holds a 0xFF byte, follows one, or holds the end of the scan can drop anything (about one
// lane in sixteen on entropy-coded data).  Those lanes are put on a list and worked off densely by the first
// lanes of the workgroup with the byte-by-byte rule below; all others keep their 16 bytes and move them with
// four funnel shifts.  (Every wave holds some such lane, so leaving the rule inside the per-lane code made
// every wave pay for it: 199 + 508 instructions per lane in round 2's first version, now about 70 + 200.)

#define UNSTUFF_T 256

struct UnstuffLds {
    uint4 raw[UNSTUFF_T];      // the chunk as loaded
    uint32_t res[UNSTUFF_T];   // listed lanes: keep mask | mark mask << 16
    uint16_t list[UNSTUFF_T];  // lanes that may drop bytes
    uint32_t pn[2];            // byte before the chunk, byte after it
    uint32_t n;                // entries of the list
    uint32_t w[17];            // block scan scratch
};

// The rule itself, for the 16 bytes of one lane: returns the keep mask, *markmask = bytes that are the second byte of a
// marker pair (restart mode).  prev = the byte before the lane's first, next = the byte after its last.
__device__ __forceinline__ uint32_t keepmask_rule(bool rst, const uint4 v, uint32_t prev, uint32_t next, int nvalid, uint32_t *markmask)
{
    if (nvalid < 16) next = 0; // the scan ends inside this lane
    const uint32_t words[4] = {v.x, v.y, v.z, v.w};
    uint32_t mask = 0, marks = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t b = (words[j >> 2] >> ((j & 3) * 8)) & 0xFF;
        uint32_t nb = j < 15 ? (words[(j + 1) >> 2] >> (((j + 1) & 3) * 8)) & 0xFF : next;
        if (j + 1 >= nvalid && nvalid < 16) nb = 0; // nothing follows the last byte of the scan
        const bool inside = j < nvalid;
        const bool second = prev == 0xFF; // the byte after an FF: a stuffing zero, or (restart mode) a marker code
        bool keep;
        if (rst) {
            const bool mfirst = b == 0xFF && nb != 0; // FF that opens a marker pair
            keep = inside && !second && !mfirst;
            marks |= ((inside && second && b != 0) ? 1u : 0u) << j;
            prev = second ? 0u : b; // a consumed pair never acts as "previous FF" (the reader restarts after it)
        } else {
            keep = inside && !(second && b == 0);
            prev = b;
        }
        mask |= (keep ? 1u : 0u) << j;
    }
    *markmask = marks;
    return mask;
}

__device__ __forceinline__ int lane_nvalid(const HjdImage &im, uint32_t lc, uint32_t lane)
{
    const int nv = (int)im.scan_len - (int)(lc * HJD_UNSTUFF_CHUNK + lane * 16u);
    return nv < 0 ? 0 : (nv > 16 ? 16 : nv);
}

// Front end of both kernels: loads the chunk, decides which lanes need the rule, applies it to those.  Returns this
// lane's keep mask | mark mask << 16 and whether it is on the list; *vout = its 16 raw bytes.
__device__ __forceinline__ uint32_t unstuff_masks(const HjdImage &im, const uint8_t *raw, uint32_t lc, UnstuffLds &L, uint4 *vout, bool *listed)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint8_t *base = raw + im.raw_off + (size_t)lc * HJD_UNSTUFF_CHUNK + tid * 16u;
    const uint4 v = *reinterpret_cast<const uint4 *>(base);
    *vout = v;
    L.raw[tid] = v;
    if (tid == 0) {
        L.pn[0] = lc == 0 ? 0u : base[-1];
        L.n = 0u;
    }
    if (tid == UNSTUFF_T - 1) L.pn[1] = base[16];
    __syncthreads();
    const uint8_t *rb = reinterpret_cast<const uint8_t *>(&L.raw[0]);
    const uint32_t prev = tid ? rb[tid * 16u - 1u] : L.pn[0];
    // any 0xFF among the 16 bytes <=> some byte of ~word is zero: (x - 0x01010101) & ~x & 0x80808080 != 0 (exact as an existence test)
    const uint32_t n0 = ~v.x, n1 = ~v.y, n2 = ~v.z, n3 = ~v.w;
    const uint32_t anyff = (((n0 - 0x01010101u) & v.x) | ((n1 - 0x01010101u) & v.y) | ((n2 - 0x01010101u) & v.z) | ((n3 - 0x01010101u) & v.w)) & 0x80808080u;
    const bool slow = anyff != 0u || prev == 0xFFu || lane_nvalid(im, lc, tid) < 16;
    const unsigned long long bal = __ballot(slow);
    uint32_t wbase = 0;
    if (lane == 0 && bal) wbase = atomicAdd(&L.n, (uint32_t)__popcll(bal));
    wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
    if (slow) L.list[wbase + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)tid;
    __syncthreads();
    const uint32_t n = L.n;
    const bool rst = im.restart != 0;
    for (uint32_t e = tid; e < n; e += UNSTUFF_T) {
        const uint32_t q = L.list[e];
        const uint32_t pq = q ? rb[q * 16u - 1u] : L.pn[0];
        const uint32_t nq = q == UNSTUFF_T - 1 ? L.pn[1] : rb[q * 16u + 16u];
        uint32_t marks;
        const uint32_t mask = keepmask_rule(rst, L.raw[q], pq, nq, lane_nvalid(im, lc, q), &marks);
        L.res[q] = mask | (marks << 16);
    }
    __syncthreads();
    *listed = slow;
    return slow ? L.res[tid] : 0xFFFFu;
}

__global__ __launch_bounds__(UNSTUFF_T) void k_unstuff_count(HjdParams P)
{
    __shared__ UnstuffLds L;
    const uint32_t c = blockIdx.x + P.off_chunk;
    const int ii = (int)P.map_chunk[c];
    const HjdImage &im = P.img[ii];
    uint32_t total; // kept bytes (<= 4096) | markers << 16 (<= 2048)
    if (!im.restart) {
        // Without DRI the rule is local — a byte is dropped iff it is 00 and the byte before it is FF (an FF is never dropped
        // itself) — so counting needs neither the list of lanes nor LDS: 0x80 per zero byte / per FF byte by the exact
        // carry-free test, the FF flags moved on by one byte through the lane's 128 bits, AND, popcount.
        const uint32_t lc = c - im.chunk0, tid = threadIdx.x;
        const uint8_t *base = P.raw + im.raw_off + (size_t)lc * HJD_UNSTUFF_CHUNK + tid * 16u;
        const uint4 q = *reinterpret_cast<const uint4 *>(base);
        const uint32_t prevb = (lc == 0u && tid == 0u) ? 0u : base[-1];
        const int nv = lane_nvalid(im, lc, tid);
        auto zero80 = [](uint32_t x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu); }; // 0x80 in every byte of x that is 0
        const uint32_t z0 = zero80(q.x), z1 = zero80(q.y), z2 = zero80(q.z), z3 = zero80(q.w);
        const uint32_t f0 = zero80(~q.x), f1 = zero80(~q.y), f2 = zero80(~q.z), f3 = zero80(~q.w);
        const uint32_t fp = prevb == 0xFFu ? 0x80u : 0u;
        // byte j of a word sits at bits 8j..8j+7: "the byte before is FF" = the FF flags shifted up by one byte, the carry coming
        // from the word (or lane) before
        uint32_t d0 = z0 & ((f0 << 8) | fp), d1 = z1 & ((f1 << 8) | (f0 >> 24)), d2 = z2 & ((f2 << 8) | (f1 >> 24)), d3 = z3 & ((f3 << 8) | (f2 >> 24));
        // bytes past the end of the scan do not count
        auto cut = [](uint32_t d, int left) { return left >= 4 ? d : (left <= 0 ? 0u : d & ((1u << (8 * left)) - 1u)); };
        d0 = cut(d0, nv), d1 = cut(d1, nv - 4), d2 = cut(d2, nv - 8), d3 = cut(d3, nv - 12);
        total = (uint32_t)nv - (uint32_t)(__popc(d0) + __popc(d1) + __popc(d2) + __popc(d3));
    } else {
        uint4 v;
        bool listed;
        const uint32_t mm = unstuff_masks(im, P.raw, c - im.chunk0, L, &v, &listed);
        total = (uint32_t)__popc(mm & 0xFFFFu) | ((uint32_t)__popc(mm >> 16) << 16);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) total += __shfl_xor(total, d, 64);
    if ((threadIdx.x & 63u) == 0u) L.w[threadIdx.x >> 6] = total; // (a sum, not a scan: one barrier)
    __syncthreads();
    total = L.w[0] + L.w[1] + L.w[2] + L.w[3];
    if (threadIdx.x == 0) {
        P.chunk_kept[c] = total & 0xFFFFu;
        P.chunk_mark[c] = total >> 16;
        if (c == im.chunk0) { // a new run of the image starts here: clear what the later kernels accumulate into
            HjdImageState z = {};
            P.st[ii] = z;
        }
    }
    if (c == im.chunk0) // ... and the final-pass workgroups' totals, which k_sync adds up
        for (uint32_t k = threadIdx.x; k < im.nwritewg; k += UNSTUFF_T) P.wgsum[im.wgsum0 + k] = make_uint2(0u, 0u);
}

__global__ __launch_bounds__(UNSTUFF_T) void k_unstuff_write(HjdParams P)
{
    __shared__ UnstuffLds L;
    __shared__ __attribute__((aligned(16))) uint8_t s_bytes[HJD_UNSTUFF_CHUNK + 32]; // the chunk's kept bytes, MSB-first words
    __shared__ uint32_t s_pos[UNSTUFF_T], s_mk[UNSTUFF_T], s_sum[2];
    const uint32_t tid = threadIdx.x;
#ifdef HJD_UNSTUFF_NO_XCD
    const uint32_t c = blockIdx.x + P.off_chunk;
#else
    // XCD-aware chunk numbering.  Workgroups are dealt to the eight XCDs round robin (workgroup b runs on XCD b mod 8), each XCD
    // has its own L2, and two neighbouring chunks write into the same sectors of the transposed stream (the rows at their common
    // border, at every word k): with chunk = workgroup number those halves of a sector sat in two L2s and reached HBM as two
    // partial writes (2.5 MB written per 4K image for 1.4 MB of stream).  Here XCD x takes a CONTIGUOUS run of chunks —
    // the first n mod 8 XCDs one more than the others — so neighbours meet in one L2.
    uint32_t c;
    {
        const uint32_t n = gridDim.x, x = blockIdx.x & 7u, j = blockIdx.x >> 3, q = n >> 3, r = n & 7u;
        c = x * q + min(x, r) + j + P.off_chunk;
    }
#endif
    const int ii = (int)P.map_chunk[c];
    const HjdImage &im = P.img[ii];
    const uint32_t lc = c - im.chunk0;
    reinterpret_cast<uint4 *>(s_bytes)[tid] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 2) reinterpret_cast<uint4 *>(s_bytes)[UNSTUFF_T + tid] = make_uint4(0u, 0u, 0u, 0u);
    // kept bytes and markers of this image's earlier chunks
    uint32_t part = 0, mpart = 0;
    for (uint32_t k = tid; k < lc; k += UNSTUFF_T) {
        part += P.chunk_kept[im.chunk0 + k];
        mpart += P.chunk_mark[im.chunk0 + k];
    }
    // their sums over the workgroup: wave sums, then two LDS atomics per wave (two block scans took six barriers for it; the
    // first barrier inside unstuff_masks orders the atomics before the reads below)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        part += __shfl_xor(part, d, 64);
        mpart += __shfl_xor(mpart, d, 64);
    }
    if (tid == 0) s_sum[0] = s_sum[1] = 0u;
    __syncthreads();
    if ((tid & 63u) == 0u) {
        atomicAdd(&s_sum[0], part);
        atomicAdd(&s_sum[1], mpart);
    }
    uint4 v;
    bool listed;
    const uint32_t mm = unstuff_masks(im, P.raw, lc, L, &v, &listed);
    const uint32_t base = s_sum[0], mbase = s_sum[1];
    uint32_t tot2;
    const uint32_t ex = block_excl_scan((uint32_t)__popc(mm & 0xFFFFu) | ((uint32_t)__popc(mm >> 16) << 16), L.w, &tot2);
    const uint32_t total = tot2 & 0xFFFFu, mtotal = tot2 >> 16;
    const uint32_t pos0 = base + (ex & 0xFFFFu);
    const uint32_t mk0 = mbase + (ex >> 16);
    // Transposed stream (see hjd_entropy.hip): word k of subsequence i -> ((i/64)*(wps+2) + k)*64 + i%64,
    // bytes MSB-first inside a word; words 0 and 1 of row i are repeated as words wps, wps+1 of row i-1.
    // The chunk's kept bytes are first compacted in LDS as MSB-first words (byte index ^ 3 for byte stores); whole
    // words then go out as 4-byte stores, the up to three bytes of the partial words at either end of the chunk
    // (shared with the neighbouring chunks) as byte stores.
    uint8_t *dst = P.ustream + im.ustream_off;
    const uint32_t lw = P.log2_words, wps = 1u << lw;
    const uint32_t wbase = base & ~3u; // stream position of LDS byte 0
    if (!listed) {
        // all 16 bytes stay: as four MSB-first words, shifted to the destination's byte phase, OR-ed into the zeroed
        // buffer (neighbouring lanes own the other bytes of the two partial words)
        const uint32_t a = pos0 - wbase, sh = (a & 3u) * 8u, q = a >> 2;
        const uint32_t b0 = __builtin_bswap32(v.x), b1 = __builtin_bswap32(v.y), b2 = __builtin_bswap32(v.z), b3 = __builtin_bswap32(v.w);
        uint32_t *sw = reinterpret_cast<uint32_t *>(s_bytes) + q;
        atomicOr(&sw[0], __builtin_amdgcn_alignbit(0u, b0, sh));
        atomicOr(&sw[1], __builtin_amdgcn_alignbit(b0, b1, sh));
        atomicOr(&sw[2], __builtin_amdgcn_alignbit(b1, b2, sh));
        atomicOr(&sw[3], __builtin_amdgcn_alignbit(b2, b3, sh));
        if (sh) atomicOr(&sw[4], b3 << (32u - sh));
    } else {
        s_pos[tid] = pos0;
        s_mk[tid] = mk0;
    }
    __syncthreads();
    for (uint32_t e = tid; e < L.n; e += UNSTUFF_T) { // listed lanes, densely: byte by byte
        const uint32_t q = L.list[e];
        const uint32_t m2 = L.res[q], mask = m2 & 0xFFFFu, marks = m2 >> 16;
        const uint4 vq = L.raw[q];
        const uint32_t words[4] = {vq.x, vq.y, vq.z, vq.w};
        uint32_t pos = s_pos[q], mk = s_mk[q];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint8_t bv = (uint8_t)(words[j >> 2] >> ((j & 3) * 8));
            if (marks & (1u << j)) { // a marker sits between kept byte pos-1 and pos
                if (mk < im.mark_cap) {
                    P.mpos[im.mark_off + mk] = pos * 8u;
                    P.mval[im.mark_off + mk] = bv;
                }
                // anything but RSTn between the intervals: the reference would take it for a resync point too,
                // with results no standard stream defines (what follows the last interval — EOI, padding —
                // is never looked at again)
                if (mk < im.n_rst && !(bv >= 0xD0 && bv <= 0xD7) && !im.conformant) atomicOr(&P.st[ii].flags, HJD_FLAG_BAD_MARKER);
                mk++;
            }
            if (mask & (1u << j)) {
                s_bytes[(pos - wbase) ^ 3u] = bv;
                pos++;
            }
        }
    }
    __syncthreads();
    auto put_word_or_byte = [&](uint32_t wq, uint32_t value, int bsel) {
        const uint32_t i = wq >> lw, k = wq & (wps - 1u);
        const size_t a0 = (((size_t)(i >> 6) * (wps + 2u) + k) * 64u + (i & 63u)) * 4u;
        if (bsel < 0) *reinterpret_cast<uint32_t *>(dst + a0) = value;
        else dst[a0 + (uint32_t)bsel] = (uint8_t)value;
        if (k < 2u && i >= 1u) {
            const uint32_t i1 = i - 1u;
            const size_t a1 = (((size_t)(i1 >> 6) * (wps + 2u) + wps + k) * 64u + (i1 & 63u)) * 4u;
            if (bsel < 0) *reinterpret_cast<uint32_t *>(dst + a1) = value;
            else dst[a1 + (uint32_t)bsel] = (uint8_t)value;
        }
    };
    {
        const uint32_t endp = base + total;                 // one past the chunk's last kept byte
        const uint32_t w_first = (base + 3u) >> 2, w_last = endp >> 2; // whole words [w_first, w_last)
#ifdef HJD_UNSTUFF_LINEAR // round 2-5 form: consecutive lanes take consecutive words of ONE row — 64 stores to 64 different 256-byte lines
        for (uint32_t wq = w_first + tid; wq < w_last; wq += UNSTUFF_T)
            put_word_or_byte(wq, *reinterpret_cast<const uint32_t *>(&s_bytes[(wq << 2) - wbase]), -1);
#else
        // Consecutive lanes take the SAME word k of consecutive rows (subsequences): in the transposed layout those are
        // neighbouring addresses, so the n rows a chunk touches leave a wave as runs of n x 4 bytes — one or two sectors — instead
        // of n lone 4-byte stores per k (a chunk of 4 KiB is 8 rows of 4096 bits, 16 of 2048).  t -> (row, k) by a multiplication
        // with the reciprocal of n (exact for t < 2^16).
        if (w_last > w_first) {
            const uint32_t ra = w_first >> lw, n = ((w_last - 1u) >> lw) - ra + 1u;
            const uint32_t rcp = n > 1u ? 0xFFFFFFFFu / n + 1u : 0u, cells = n << lw;
            for (uint32_t t = tid; t < cells; t += UNSTUFF_T) {
                const uint32_t k = n > 1u ? __umulhi(t, rcp) : t, row = ra + (t - k * n);
                const uint32_t wq = (row << lw) + k;
                if (wq >= w_first && wq < w_last) put_word_or_byte(wq, *reinterpret_cast<const uint32_t *>(&s_bytes[(wq << 2) - wbase]), -1);
            }
        }
#endif
        // bytes of the partial words: head = [base, min(4*w_first, endp)), tail = [4*w_last, endp) when w_last >= w_first
        if (tid < 4) {
            const uint32_t q = base + tid;
            if (q < min(w_first << 2, endp)) put_word_or_byte(q >> 2, s_bytes[(q - wbase) ^ 3u], (int)(3u - (q & 3u)));
        } else if (tid < 8 && w_last >= w_first) {
            const uint32_t q = (w_last << 2) + (tid - 4);
            if (q < endp) put_word_or_byte(q >> 2, s_bytes[(q - wbase) ^ 3u], (int)(3u - (q & 3u)));
        }
    }
    if (lc == im.nchunks - 1 && tid == 0) {
        P.st[ii].ustream_len = base + total;
        P.st[ii].nmarkers = mbase + mtotal;
    }
}

// ---- stage 3b: DC prediction --------------------------------------------------------------------
// DCT[0] = diff + previousDC per component, predictors zeroed once per image and never reset
// (loadjpg.cpp:619-667, :1159-1162), int16 wrap (loadjpg.cpp:664-665) = prefix sums mod 2^16 over the
// blocks of each component in decode order.  Two passes over chunks of HJD_DC_CHUNK MCUs:
// k_dcsum (per-chunk totals), k_dcapply (totals of the earlier chunks + scan inside the chunk).
// The differences sit in the blocks' 8-byte entries (blk[].y, written by k_write together with the first-record index: one store
// per block); the predicted values go to the compact array dca[] (a thread's 2-byte stores are contiguous: whole sectors
// leave the wave).

#define DC_T (HJD_DC_CHUNK / 256) // MCUs per thread

__device__ __forceinline__ uint32_t dc_diff(const uint2 &e) { return (uint32_t)(int)(int16_t)e.y; }

__device__ __forceinline__ void dc_thread_sums(const uint2 *d, uint32_t m0, uint32_t m1, uint32_t nb, uint32_t nby,
                                               uint32_t &y, uint32_t &cb, uint32_t &cr)
{
    y = cb = cr = 0;
    for (uint32_t mc = m0; mc < m1; mc++) {
        const uint2 *q = d + (size_t)mc * nb;
        for (uint32_t k = 0; k < nby; k++) y += dc_diff(q[k]);
        cb += dc_diff(q[nby]);
        cr += dc_diff(q[nby + 1]);
    }
}

// Three sums at once over a workgroup of 256 lanes, ONE barrier each (three block_excl_scan calls were nine; these two kernels are
// nothing but latency).  s: 12 words of LDS that nothing else touches until the next barrier.
__device__ __forceinline__ void block_sum3(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t *s)
{
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        a += __shfl_xor(a, d, 64);
        b += __shfl_xor(b, d, 64);
        c += __shfl_xor(c, d, 64);
    }
    if (lane == 0) s[wv * 3u] = a, s[wv * 3u + 1u] = b, s[wv * 3u + 2u] = c;
    __syncthreads();
    a = s[0] + s[3] + s[6] + s[9];
    b = s[1] + s[4] + s[7] + s[10];
    c = s[2] + s[5] + s[8] + s[11];
}

// exclusive prefixes of three values over the workgroup's lanes, in place
__device__ __forceinline__ void block_excl_scan3(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t *s)
{
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t ia = a, ib = b, ic = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t ta = __shfl_up(ia, d, 64), tb = __shfl_up(ib, d, 64), tc = __shfl_up(ic, d, 64);
        if ((int)lane >= d) ia += ta, ib += tb, ic += tc;
    }
    if (lane == 63u) s[wv * 3u] = ia, s[wv * 3u + 1u] = ib, s[wv * 3u + 2u] = ic;
    __syncthreads();
    uint32_t pa = 0, pb = 0, pc = 0;
    for (uint32_t k = 0; k < wv; k++) pa += s[k * 3u], pb += s[k * 3u + 1u], pc += s[k * 3u + 2u];
    a = pa + ia - a;
    b = pb + ib - b;
    c = pc + ic - c;
}

__global__ __launch_bounds__(256) void k_dcsum(HjdParams P)
{
    __shared__ uint32_t s_w[12];
    const uint32_t bid = blockIdx.x + P.off_dcchunk;
    const int ii = (int)P.map_dcchunk[bid];
    const HjdImage &im = P.img[ii];
    const uint32_t nmcu = im.mcux * im.mcuy;
    const uint32_t c0 = (bid - im.dcchunk0) * HJD_DC_CHUNK;
    const uint32_t m0 = min(c0 + threadIdx.x * DC_T, nmcu), m1 = min(m0 + DC_T, nmcu);
    uint32_t y, cb, cr;
    dc_thread_sums(P.blk + im.blk_off, m0, m1, im.nb, im.nby, y, cb, cr);
    block_sum3(y, cb, cr, s_w);
    if (threadIdx.x == 0) {
        P.dcsum[bid * 3 + 0] = y;
        P.dcsum[bid * 3 + 1] = cb;
        P.dcsum[bid * 3 + 2] = cr;
    }
}

__global__ __launch_bounds__(256) void k_dcapply(HjdParams P)
{
    __shared__ uint32_t s_a[12], s_b[12];
    const uint32_t bid = blockIdx.x + P.off_dcchunk;
    const int ii = (int)P.map_dcchunk[bid];
    const HjdImage &im = P.img[ii];
    const uint32_t nb = im.nb, nby = im.nby;
    const uint32_t nmcu = im.mcux * im.mcuy;
    const uint32_t lc = bid - im.dcchunk0;
    // totals of this image's earlier chunks
    uint32_t by = 0, bcb = 0, bcr = 0;
    for (uint32_t k = threadIdx.x; k < lc; k += blockDim.x) {
        by += P.dcsum[(im.dcchunk0 + k) * 3 + 0];
        bcb += P.dcsum[(im.dcchunk0 + k) * 3 + 1];
        bcr += P.dcsum[(im.dcchunk0 + k) * 3 + 2];
    }
    const uint32_t m0 = min(lc * HJD_DC_CHUNK + threadIdx.x * DC_T, nmcu), m1 = min(m0 + DC_T, nmcu);
    const uint2 *d = P.blk + im.blk_off;
    int16_t *o = P.dca + im.blk_off;
    uint32_t y, cb, cr;
    dc_thread_sums(d, m0, m1, nb, nby, y, cb, cr); // (its loads are in flight while the totals are added up)
    block_sum3(by, bcb, bcr, s_a);
    block_excl_scan3(y, cb, cr, s_b);
    y += by;
    cb += bcb;
    cr += bcr;
    for (uint32_t mc = m0; mc < m1; mc++) {
        const uint2 *q = d + (size_t)mc * nb;
        int16_t *w = o + (size_t)mc * nb;
        for (uint32_t k = 0; k < nby; k++) {
            y += dc_diff(q[k]);
            w[k] = (int16_t)y; // int16 wrap of the predictor (loadjpg.cpp:664-665)
        }
        cb += dc_diff(q[nby]);
        w[nby] = (int16_t)cb;
        cr += dc_diff(q[nby + 1]);
        w[nby + 1] = (int16_t)cr;
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

int hjd_launch_dcscan(const HjdParams &P, uint32_t nchunks, hipStream_t s)
{
    if (!nchunks) return 0;
    hipLaunchKernelGGL(k_dcsum, dim3(nchunks), dim3(256), 0, s, P);
    HJD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_dcapply, dim3(nchunks), dim3(256), 0, s, P);
    HJD_LAUNCH_CHECK();
    return 0;
}
