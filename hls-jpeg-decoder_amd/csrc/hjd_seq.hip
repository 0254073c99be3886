// hjd_seq.hip — the reference's entropy decoder, literally and sequentially (gfx950), for the restart
// streams on which it leaves the clean-resync domain.
//   k_seq_decode   ProcessHuffmanBlock + FillNBits with the reference's BYTE-CURSOR rules
//                  (loadjpg.cpp:446-484, :499-863, :945-997, :1148-1182)
//
// The parallel path (hjd_entropy.hip) is valid where "jump to the marker at the end of an interval" equals
// what the reference does.  Where it does not — the resync check fires on a block whose real bits are
// still in the reservoir (:535-550), or between a DC code and its extra bits (:631-640), or a marker is
// swallowed as data by the bit reader (:468-482) — the reference still produces defined bytes.  This kernel
// reproduces them: ONE lane per image walks the stuffed scan exactly as the reference does (byte cursor,
// 32-bit reservoir filled at most two bytes at a time, resync = cursor += 2 and reservoir cleared, DC
// predictors never reset) and writes the same outputs as k_write + the DC scan: records, block entries,
// predicted DC.  It is slow (one dependent chain per image) and only launched for images the parallel path
// flagged DIRTY.  Like the reference (and the oracle) it reads zeros past the end of the scan: a desynchronised
// decode may well finish its blocks there.  What the reference cannot decode either — no matching code (it loops forever,
// :700-829), a zig-zag index past 63 (it writes outside DCT[], :780-806) — is flagged, not emulated.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hjd_device.h"
#include "hjd_kernels.h"

__constant__ uint8_t c_zz_to_slot_seq[64] = HJD_ZZ_TO_SLOT;

struct SeqReader { // stHuffmanData::m_stream / stream_index / g_reservoir / g_nbits_in_reservoir (loadjpg.h:149-153)
    const uint8_t *s;
    uint32_t n, idx, res, nbits;
    // The scan is read through a 32-byte register window (two 16-byte loads, the second fetched 16 bytes before it is needed)
    // instead of one dependent global load per byte: the lane's chain was mostly load latency (1080p: 120 ms).
    uint32_t b0 = 0xFFFFFFE0u; // bytes [b0, b0 + 16) in w0, [b0 + 16, b0 + 32) in w1; the scan's slab is 16-byte aligned and padded
    uint4 w0, w1;
    __device__ __forceinline__ uint4 load16(uint32_t base) const
    {
        return base < n ? *reinterpret_cast<const uint4 *>(s + base) : make_uint4(0u, 0u, 0u, 0u);
    }
    __device__ __forceinline__ uint32_t at(uint32_t i)
    {
        if (i >= n) return 0u;
        uint32_t q = i - b0;
        if (q >= 32u) { // first access, or a resynchronisation jumped ahead
            b0 = i & ~15u;
            w0 = load16(b0);
            w1 = load16(b0 + 16u);
            q = i - b0;
        } else if (q >= 24u) { // slide: keeps eight bytes behind the cursor (the look-ahead is one byte)
            b0 += 16u;
            w0 = w1;
            w1 = load16(b0 + 16u);
            q -= 16u;
        }
        const uint4 w = q < 16u ? w0 : w1;
        const uint32_t k = (q >> 2) & 3u;
        const uint32_t word = k == 0u ? w.x : (k == 1u ? w.y : (k == 2u ? w.z : w.w));
        return (word >> ((q & 3u) * 8u)) & 0xFFu;
    }
    __device__ __forceinline__ void fill(uint32_t limit) // FillNBits, live branch loadjpg.cpp:468-482
    {
        for (int k = 0; k < 2; k++)
            if (nbits < limit) {
                const uint32_t c = at(idx);
                idx++;
                res <<= 8;
                if (c == 0xFFu && at(idx) == 0x00u) idx++;
                res |= c;
                nbits += 8;
            }
    }
    __device__ __forceinline__ void skip(uint32_t k) // loadjpg.cpp:607-609
    {
        fill(k);
        nbits -= k;
        res &= (1u << nbits) - 1u;
    }
    __device__ __forceinline__ uint32_t get(uint32_t k) // :647-652
    {
        fill(k);
        const uint32_t v = res >> (nbits - k);
        nbits -= k;
        res &= (1u << nbits) - 1u;
        return v;
    }
    __device__ __forceinline__ bool on_marker() { return at(idx) == 0xFFu && at(idx + 1) != 0x00u; }
    __device__ __forceinline__ void resync() // :544-546 / :634-636
    {
        idx += 2;
        res = 0;
        nbits = 0;
    }
};

__device__ __forceinline__ int seq_extend(uint32_t v, uint32_t n) // DetermineSign, loadjpg.cpp:396-409
{
    return v < (1u << (n - 1u)) ? (int)v - (int)((1u << n) - 1u) : (int)v;
}

// One workgroup per image with DRI; it returns at once unless the parallel path has flagged the image (the launch is part of
// every run of a batch that holds such images, on a side stream: no host round trip, no allocation — hjd_api.cpp).
__global__ __launch_bounds__(64) void k_seq_decode(HjdParams P, uint32_t first)
{
    // The reference tries the code lengths one after the other (IsInHuffmanCodes over k = 1..16, loadjpg.cpp:562, :713); for the
    // prefix-free tables the parser admits that finds the same (length, symbol) as one look-up of the next 16 bits in the
    // decode tables of the parallel path, and the k FillNBits calls of the search leave the reader where one FillNBits(length)
    // leaves it (each loads at most the one byte that is missing).  So: peek 16 bits on a COPY of the reader, look the code up
    // (11-bit primary table in LDS, secondary in HBM), then consume it with the reader proper.  (Length by length out of LDS
    // this kernel spent 0.9 us per symbol; it is a one-lane chain, so that is its speed.)
    __shared__ uint16_t s_lut[HJD_SLOTS][HJD_LUT_SIZE];
    const int ii = (int)P.rst_list[first + blockIdx.x];
    if (!(P.st[ii].flags & (HJD_FLAG_DIRTY | HJD_FLAG_BAD_MARKER))) return; // whatever else the parallel path flagged after leaving
                                                                            // the clean-resync domain means nothing
    const HjdImage &im = P.img[ii];
    const HjdTableSet &ts = P.tabs[im.tabset];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(&ts.lut[0][0]);
        uint4 *dst = reinterpret_cast<uint4 *>(&s_lut[0][0]);
        for (uint32_t k = threadIdx.x; k < ts.nlut * (HJD_LUT_SIZE * 2u / 16u); k += 64u) dst[k] = src[k];
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    // (length, value bits, zig-zag advance) of the code at the head of the reader's bits, table `ti`; false = no such code
    auto peek_code = [&](const SeqReader &r0, uint32_t ti, uint32_t &len, uint32_t &size, uint32_t &adv) -> bool {
        SeqReader t = r0;
        t.fill(16u); // at most two bytes: 16 bits are there afterwards
        const uint32_t c16 = (t.res >> (t.nbits - 16u)) & 0xFFFFu;
        uint32_t e = s_lut[ti][c16 >> (16 - HJD_LUT_BITS)];
        if ((e & 31u) == 0u) {
            const uint32_t sub = e >> 5;
            e = sub ? ts.sec[sub - 1u][c16 & (HJD_SEC_SIZE - 1u)] : 0u;
        }
        len = e & 31u;
        size = (e >> 5) & 15u;
        adv = e >> 9;
        return len != 0u;
    };
    SeqReader r;
    r.s = P.raw + im.raw_off;
    r.n = im.scan_len;
    r.idx = r.res = r.nbits = 0u; // loadjpg.cpp:1148-1149
    uint32_t *rec = P.rec + im.rec_off;
    uint2 *blk = P.blk + im.blk_off;
    int16_t *dca = P.dca + im.blk_off;
    int prev[3] = {0, 0, 0}; // :1159-1162, never reset afterwards
    uint32_t nrec = 0, flags = 0;
    const bool restart = im.restart != 0;

    for (uint32_t b = 0; b < im.nblocks && !flags; b++) {
        const uint32_t m = b % im.nb;
        const uint32_t comp = m < im.nby ? 0u : m - im.nby + 1u; // DecodeMCU order, :945-997
        const uint32_t tdc = (ts.slot_pack >> (comp * 8u)) & 7u, tac = (ts.slot_pack >> (comp * 8u + 4u)) & 7u;
        const uint32_t first = nrec;

        if (restart && r.on_marker()) r.resync(); // block-start check, :535-550

        int dc = prev[comp];
        {
            uint32_t len, size, adv;
            // :562: DC codes of 1..15 bits.  No match: the reference prints and carries on with the AC loop on the same bits
            // (:680-685): not emulated
            if (!peek_code(r, tdc, len, size, adv) || len > 15u) {
                flags |= HJD_FLAG_BAD_CODE;
                break;
            }
            r.skip(len);
            if (size != 0u) {
                if (restart && r.on_marker()) r.resync(); // between the code and its extra bits, :631-640
                const int data = (int)(int16_t)seq_extend(r.get(size), size);
                dc = (int)(int16_t)(data + prev[comp]); // :664
                prev[comp] = dc;
            }
        }

        uint32_t nr = 1;
        bool eob = false;
        while (nr <= 63u && !eob) { // :700
            uint32_t len, size, adv;
            if (!peek_code(r, tac, len, size, adv)) { // no code of 1..16 bits: the reference's loop makes no progress
                flags |= HJD_FLAG_BAD_CODE;
                break;
            }
            r.skip(len);
            if (size == 0u) { // :768-775: adv = 64 end of block, 16 ZRL, 0 any other run with size 0 (ignored)
                if (adv == 64u) eob = true;
                else if (adv == 16u) nr += 16u;
                else if (r.idx > im.scan_len + 4u) flags |= HJD_FLAG_BAD_CODE; // an ignored symbol decoded from the zeros past
                                                                               // the end: the reference repeats it forever
            } else {
                nr += adv - 1u; // run, :778
                if (nr > 63u) { // the reference writes DCT[nr] out of bounds here (:780-806)
                    flags |= HJD_FLAG_OVERRUN;
                    break;
                }
                const int data = (int)(int16_t)seq_extend(r.get(size), size);
                const uint32_t slot = c_zz_to_slot_seq[nr];
                if (nrec < im.rec_cap) rec[nrec] = ((uint32_t)data & 0xFFFFu) | (slot << 16) | ((b & 127u) << 22);
                else flags |= HJD_FLAG_OVERRUN;
                nrec++;
                nr++;
            }
            if (flags) break;
        }
        blk[b] = make_uint2(first, 0u); // the back end takes the predicted DC from dca[], the slots its records fill from the records
        dca[b] = (int16_t)dc;
    }
    blk[im.nblocks] = make_uint2(min(nrec, im.rec_cap), 0u);
    HjdImageState &st = P.st[ii];
    // replaces the parallel path's verdict for this image; what the unstuff stage found (BAD_MARKER) stays, because a host
    // re-synchronisation re-runs everything after that stage and must send the image here again
    st.flags = flags | (st.flags & HJD_FLAG_BAD_MARKER);
    st.total_blocks = flags ? 0u : im.nblocks;
    st.emulated = 1u;
}

int hjd_launch_seq_decode(const HjdParams &P, uint32_t first, uint32_t n, hipStream_t s)
{
    if (!n) return 0;
    hipLaunchKernelGGL(k_seq_decode, dim3(n), dim3(64), 0, s, P, first);
    hipError_t e_ = hipGetLastError();
    return e_ == hipSuccess ? 0 : (int)e_;
}
