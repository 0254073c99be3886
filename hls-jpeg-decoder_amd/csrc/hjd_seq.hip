// hjd_seq.hip — the reference's entropy decoder, literally and sequentially (gfx950), for the restart
// streams on which it leaves the clean-resync domain.
//   k_seq_decode   ProcessHuffmanBlock + FillNBits with the reference's BYTE-CURSOR rules
//                  (loadjpg.cpp:446-484, :499-863, :945-997, :1148-1182), one lane per image
//   k_seg_*        the same decode, parallel over the places where the reference can resynchronise (below)
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
#include "hjd_kutil.h"

__constant__ uint8_t c_zz_to_slot_seq[64] = HJD_ZZ_TO_SLOT;

struct SeqReader { // stHuffmanData::m_stream / stream_index / g_reservoir / g_nbits_in_reservoir (loadjpg.h:149-153)
    const uint8_t *s;
    uint32_t n, idx, res, nbits;
    // The scan is read through an eight-byte register window (three aligned 4-byte loads and two v_alignbyte per refill, a refill
    // every six or seven bytes) instead of one dependent global load per byte.  (Round 3: a 32-byte window of two uint4 — the
    // byte select through four words and the slide cost some 20 instructions per byte; the emulation kernels are instruction-bound.)
    uint64_t w = 0;                // raw bytes [wpos, wpos + 8), byte k at bits 8k..8k+7; bytes past the end of the scan are 00
    uint32_t wpos = 0xFFFFFFF0u;
    __device__ __forceinline__ void refill(uint32_t pos)
    {
        uint64_t v = 0;
        if (pos < n) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(s + (pos & ~3u)); // the slab is 16-byte aligned and padded by 16 bytes
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], sh = pos & 3u;
            v = ((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32) | __builtin_amdgcn_alignbyte(d1, d0, sh);
            const uint32_t left = n - pos;
            if (left < 8u) v &= (1ull << (8u * left)) - 1ull;
        }
        w = v;
        wpos = pos;
    }
    __device__ __forceinline__ uint32_t at(uint32_t i) // i is idx or idx + 1
    {
        uint32_t k = i - wpos;
        if (k >= 8u) { // first access, the cursor has left the window, or a resynchronisation jumped ahead
            refill(min(i, idx));
            k = i - wpos;
        }
        return (uint32_t)(w >> (k * 8u)) & 0xFFu;
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

typedef uint16_t SeqLut[HJD_LUT_SIZE];

// The image's primary decode tables into LDS (blockDim.x threads; the caller puts a barrier behind it).
__device__ __forceinline__ void seq_load_lut(SeqLut *s_lut, const HjdTableSet &ts)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(&ts.lut[0][0]);
    uint4 *dst = reinterpret_cast<uint4 *>(&s_lut[0][0]);
    for (uint32_t k = threadIdx.x; k < ts.nlut * (HJD_LUT_SIZE * 2u / 16u); k += blockDim.x) dst[k] = src[k];
}

// The reference tries the code lengths one after the other (IsInHuffmanCodes over k = 1..16, loadjpg.cpp:562, :713); for the
// prefix-free tables the parser admits that finds the same (length, symbol) as one look-up of the next 16 bits in the
// decode tables of the parallel path, and the k FillNBits calls of the search leave the reader where one FillNBits(length)
// leaves it (each loads at most the one byte that is missing).  So: peek 16 bits on a COPY of the reader, look the code up
// (11-bit primary table in LDS, secondary in HBM), then consume it with the reader proper.  (Length by length out of LDS
// the sequential kernel spent 0.9 us per symbol; it is a one-lane chain, so that is its speed.)
// (length, value bits, zig-zag advance) of the code at the head of the reader's bits, table `ti`; false = no such code
__device__ __forceinline__ bool seq_peek(const SeqReader &r0, const SeqLut *s_lut, const HjdTableSet &ts, uint32_t ti, uint32_t &len, uint32_t &size,
                                         uint32_t &adv)
{
    SeqReader t = r0;
    t.fill(16u); // at most two bytes: 16 bits are there afterwards
    const uint32_t c16 = (t.res >> (t.nbits - 16u)) & 0xFFFFu;
    uint32_t e = s_lut[ti][c16 >> (16 - HJD_LUT_BITS)];
    if ((e & 31u) == 0u) {
        const uint32_t sub = e >> 5;
        e = sub ? ts.sec[sub - 1u][c16 & (HJD_SEC_SIZE - 1u)] : 0u;
    }
    size = (e >> 5) & 15u; // e = code length + value bits | value bits << 5 | advance << 9
    len = (e & 31u) - size;
    adv = e >> 9;
    return (e & 31u) != 0u;
}

// ONE block as ProcessHuffmanBlock decodes it (loadjpg.cpp:499-863) behind its block-start check, which is the caller's:
// returns HJD_FLAG_* (0 = decoded).  *diff = the DC difference (0 for a code without extra bits: the predictor stays, :664);
// nrec counts the block's records on; WRITE stores them (block number b in the record, as k_write does).
template <bool WRITE>
__device__ __forceinline__ uint32_t seq_block(SeqReader &r, const SeqLut *s_lut, const HjdTableSet &ts, const HjdImage &im, bool restart, uint32_t comp,
                                              uint32_t b, int *diff, uint32_t &nrec, uint32_t *rec)
{
    const uint32_t tdc = (ts.slot_pack >> (comp * 8u)) & 7u, tac = (ts.slot_pack >> (comp * 8u + 4u)) & 7u;
    *diff = 0;
    {
        uint32_t len, size, adv;
        // :562: DC codes of 1..15 bits.  No match: the reference prints and carries on with the AC loop on the same bits
        // (:680-685): not emulated
        if (!seq_peek(r, s_lut, ts, tdc, len, size, adv) || len > 15u) return HJD_FLAG_BAD_CODE;
        r.skip(len);
        if (size != 0u) {
            if (restart && r.on_marker()) r.resync(); // between the code and its extra bits, :631-640
            *diff = (int)(int16_t)seq_extend(r.get(size), size);
        }
    }
    uint32_t nr = 1;
    bool eob = false;
    while (nr <= 63u && !eob) { // :700
        uint32_t len, size, adv;
        if (!seq_peek(r, s_lut, ts, tac, len, size, adv)) return HJD_FLAG_BAD_CODE; // no code of 1..16 bits: the reference's loop makes no progress
        r.skip(len);
        if (size == 0u) { // :768-775: adv = 64 end of block, 16 ZRL, 0 any other run with size 0 (ignored)
            if (adv == 64u) eob = true;
            else if (adv == 16u) nr += 16u;
            else if (r.idx > im.scan_len + 4u) return HJD_FLAG_BAD_CODE; // an ignored symbol decoded from the zeros past the end: the
                                                                         // reference repeats it forever
        } else {
            nr += adv - 1u; // run, :778
            if (nr > 63u) return HJD_FLAG_OVERRUN; // the reference writes DCT[nr] out of bounds here (:780-806)
            const int data = (int)(int16_t)seq_extend(r.get(size), size);
            if (WRITE) {
                if (nrec >= im.rec_cap) return HJD_FLAG_OVERRUN;
                rec[nrec] = ((uint32_t)data & 0xFFFFu) | ((uint32_t)c_zz_to_slot_seq[nr] << 16) | ((b & 127u) << 22);
            }
            nrec++;
            nr++;
        }
    }
    return 0u;
}

__device__ __forceinline__ uint32_t seq_comp(const HjdImage &im, uint32_t m) { return m < im.nby ? 0u : m - im.nby + 1u; } // DecodeMCU order, :945-997

// One workgroup per image with DRI; it returns at once unless the parallel path has flagged the image (the launch is part of
// every run of a batch that holds such images, on a side stream: no host round trip, no allocation — hjd_api.cpp).
// after_seg: launched behind the segment-parallel emulation below — what that one has decoded is left alone.
__global__ __launch_bounds__(64) void k_seq_decode(HjdParams P, uint32_t first, uint32_t after_seg)
{
    __shared__ SeqLut s_lut[HJD_SLOTS];
    const int ii = (int)P.rst_list[first + blockIdx.x];
    if (!(P.st[ii].flags & (HJD_FLAG_DIRTY | HJD_FLAG_BAD_MARKER))) return; // whatever else the parallel path flagged after leaving
                                                                            // the clean-resync domain means nothing
    if (after_seg && P.st[ii].emulated) return;
    const HjdImage &im = P.img[ii];
    const HjdTableSet &ts = P.tabs[im.tabset];
    seq_load_lut(s_lut, ts);
    __syncthreads();
    if (threadIdx.x != 0) return;
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
        const uint32_t comp = seq_comp(im, b % im.nb);
        const uint32_t first_rec = nrec;
        if (restart && r.on_marker()) r.resync(); // block-start check, :535-550
        int diff;
        flags |= seq_block<true>(r, s_lut, ts, im, restart, comp, b, &diff, nrec, rec);
        if (flags) break;
        prev[comp] = (int)(int16_t)(diff + prev[comp]); // :664
        blk[b] = make_uint2(first_rec, 0u); // the back end takes the predicted DC from dca[], the slots its records fill from the records
        dca[b] = (int16_t)prev[comp];
    }
    blk[im.nblocks] = make_uint2(min(nrec, im.rec_cap), 0u);
    HjdImageState &st = P.st[ii];
    // replaces the parallel path's verdict for this image; what the unstuff stage found (BAD_MARKER) stays, because a host
    // re-synchronisation re-runs everything after that stage and must send the image here again
    st.flags = flags | (st.flags & HJD_FLAG_BAD_MARKER);
    st.total_blocks = flags ? 0u : im.nblocks;
    st.emulated = 1u;
}

// ---- the same emulation, parallel over the resynchronisations ----------------------------------------------------------------
// After a resync at a block start the reference's reader is in a KNOWN state: cursor = marker + 2, reservoir empty
// (loadjpg.cpp:544-546).  What is not known without the decode up to there is the block it believes it is in.  So every
// place the check can fire — a raw byte FF followed by a byte that is not 00: a CANDIDATE — is a segment start, and a segment is
// decoded under each of the nb block-in-MCU labels by a lane of its own (node = candidate x label; node 0 = the start of the
// scan, label 0) until ITS block-start check fires: that names the next candidate, the label there is (label + blocks) mod nb.
// The true decode is the chain of nodes from node 0; one lane per image follows it (out of LDS) and adds up blocks, records and
// DC differences, then the chain's nodes are decoded once more, in parallel, this time writing records, block entries and
// predicted DC at their places.  A resync between a DC code and its extra bits (:631-640) does not end a segment: the lane
// carries on through it as the reference does.  Nodes are cut off after seg_cap_blocks() blocks (a wrong label may run through
// many markers); when the TRUE chain meets a cut-off node, or the candidates outnumber the space reserved, the image is left
// to k_seq_decode.
#define SEG_RESYNC 0u
#define SEG_ERR 1u
#define SEG_EOD 2u
#define SEG_CUT 3u
#define SEG_DEAD 4u
#define SEG_NONE 0xFFFFFFFFu
#define SEG_TILE 3072u // nodes of a chain tile in LDS
#define SEG_WG 256u    // lanes (= nodes of a work item) of the two persistent kernels: four waves share the 24 KB of tables

__device__ __forceinline__ uint32_t seg_cap_blocks(const HjdParams &P, const HjdImage &im)
{
    return (P.seg_cut & 0x80000000u) ? (P.seg_cut & 0x7FFFFFFFu) : max(P.seg_cut, 2u * im.seg_ri + 64u); // top bit: the test knob's value as it is
}

// Exclusive prefix of the images' work items (SEG_WG nodes each): which 0 = segments (all nodes), 1 = the write pass (the chain's
// nodes).  Run by the LAST workgroup of k_seg_list / k_seg_chain to finish (a counter behind the plans; it leaves it at 0 for
// the next run): a launch less each.  All lanes of that workgroup call it; s_w: 17 words.
__device__ __forceinline__ void seg_plan(const HjdParams &P, uint32_t which, uint32_t *s_w)
{
    uint32_t *plan = P.seg_plan + (size_t)which * (P.nrst + 1u);
    uint32_t run = 0;
    for (uint32_t q0 = 0; q0 < P.nrst; q0 += blockDim.x) { // uniform
        const uint32_t q = q0 + threadIdx.x;
        uint32_t items = 0;
        if (q < P.nrst) {
            if (which == 0u) {
                const uint32_t c = __hip_atomic_load(&P.seg_cnt[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (c != SEG_NONE) items = ((c + 1u) * P.img[P.rst_list[q]].nb + SEG_WG - 1u) / SEG_WG;
            } else items = (__hip_atomic_load(&P.seg_nchain[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + SEG_WG - 1u) / SEG_WG;
        }
        uint32_t sum;
        const uint32_t ex = run + block_excl_scan(items, s_w, &sum);
        if (q < P.nrst) plan[q] = ex;
        run += sum;
    }
    if (threadIdx.x == 0) plan[P.nrst] = run;
}

// true in every lane of the last of the grid's workgroups to get here (everything the others wrote before is visible to it)
__device__ __forceinline__ bool seg_last_workgroup(const HjdParams &P, uint32_t which, uint32_t *s_flag)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t *ctr = P.seg_plan + 2u * (P.nrst + 1u) + which;
        __threadfence();
        const uint32_t old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        *s_flag = old == gridDim.x - 1u ? 1u : 0u;
        if (old == gridDim.x - 1u) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return *s_flag != 0u;
}

// candidates of one image, ascending; grid = the images with DRI
__global__ __launch_bounds__(1024) void k_seg_list(HjdParams P)
{
    __shared__ uint32_t s_w[17], s_flag;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const int ii = (int)P.rst_list[q];
    const HjdImage &im = P.img[ii];
    const bool want = (P.st[ii].flags & (HJD_FLAG_DIRTY | HJD_FLAG_BAD_MARKER)) && im.seg_cap != 0u;
    if (!want) {
        if (tid == 0) P.seg_cnt[q] = SEG_NONE, P.seg_nchain[q] = 0u;
        if (seg_last_workgroup(P, 0u, &s_flag)) seg_plan(P, 0u, s_w);
        return;
    }
    if (tid == 0) P.st[ii].emulated = 0u; // a host re-synchronisation runs this again
    const uint8_t *raw = P.raw + im.raw_off;
    const uint32_t n = im.scan_len;
    uint32_t *pos = P.seg_pos + im.seg_off;
    uint32_t total = 0;
    for (uint32_t base = 0; base < n; base += 1024u * 64u) { // uniform; 64 bytes a lane, 1024 lanes: one block scan per 64 KB
        const uint32_t o = base + tid * 64u;
        uint32_t w[17];
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (o + k * 16u < n) v = *reinterpret_cast<const uint4 *>(raw + o + k * 16u); // the slab is 16-byte aligned and padded
            w[k * 4u] = v.x, w[k * 4u + 1u] = v.y, w[k * 4u + 2u] = v.z, w[k * 4u + 3u] = v.w;
        }
        w[16] = o + 64u < n ? raw[o + 64u] : 0u;
        uint64_t mask = 0;
#pragma unroll
        for (uint32_t k = 0; k < 64u; k++) {
            const uint32_t c = (w[k >> 2] >> ((k & 3u) * 8u)) & 0xFFu, d = (w[(k + 1u) >> 2] >> (((k + 1u) & 3u) * 8u)) & 0xFFu;
            if (c == 0xFFu && d != 0u && o + k + 1u < n) mask |= 1ull << k; // SeqReader::on_marker: bytes past the end read as 00
        }
        uint32_t sum;
        uint32_t at = total + block_excl_scan((uint32_t)__popcll(mask), s_w, &sum);
        while (mask) {
            const uint32_t k = (uint32_t)__ffsll((long long)mask) - 1u;
            mask &= mask - 1ull;
            if (at < im.seg_cap) pos[at] = o + k;
            at++;
        }
        total += sum;
    }
    if (tid == 0) P.seg_cnt[q] = total <= im.seg_cap ? total : SEG_NONE, P.seg_nchain[q] = 0u;
    if (seg_last_workgroup(P, 0u, &s_flag)) seg_plan(P, 0u, s_w);
}

// work item -> (entry of rst_list, chunk of SEG_WG nodes): the last q with plan[q] <= item
__device__ __forceinline__ uint32_t seg_find(const uint32_t *plan, uint32_t nrst, uint32_t item)
{
    uint32_t lo = 0, hi = nrst - 1u;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1u) >> 1;
        if (plan[mid] <= item) lo = mid;
        else hi = mid - 1u;
    }
    return lo;
}

__device__ __forceinline__ void seg_start(SeqReader &r, const HjdParams &P, const HjdImage &im, uint32_t j)
{
    r.s = P.raw + im.raw_off;
    r.n = im.scan_len;
    r.idx = j ? P.seg_pos[im.seg_off + (j - 1u)] + 2u : 0u; // behind the resync at candidate j - 1; node 0: the start of the scan
    r.res = r.nbits = 0u;
}

// every node's segment, counted: a persistent grid over the work items of seg_plan(0) (none: every workgroup returns at once)
__global__ __launch_bounds__(SEG_WG) void k_seg_decode(HjdParams P)
{
    __shared__ SeqLut s_lut[HJD_SLOTS];
    const uint32_t *plan = P.seg_plan;
    const uint32_t total = plan[P.nrst];
    uint32_t have = 0xFFFFFFFFu; // table set in LDS
    for (uint32_t item = blockIdx.x; item < total; item += gridDim.x) { // uniform
        const uint32_t q = seg_find(plan, P.nrst, item);
        const int ii = (int)P.rst_list[q];
        const HjdImage &im = P.img[ii];
        const HjdTableSet &ts = P.tabs[im.tabset];
        if (have != im.tabset) {
            __syncthreads();
            seq_load_lut(s_lut, ts);
            have = im.tabset;
            __syncthreads();
        }
        const uint32_t cnt = P.seg_cnt[q], node = (item - plan[q]) * SEG_WG + threadIdx.x;
        if (node >= (cnt + 1u) * im.nb) continue;
        const uint32_t j = node / im.nb, m0 = node % im.nb;
        uint32_t kind = SEG_DEAD, nblk = 0, nrec = 0, dc[3] = {0u, 0u, 0u}, nextpos = 0;
        if (j != 0u || m0 == 0u) {
            SeqReader r;
            seg_start(r, P, im, j);
            bool check = j == 0u; // a node behind a resync starts behind the check (the reference checks once per block, :535)
            const uint32_t cap = seg_cap_blocks(P, im);
            for (;;) {
                if (check && r.on_marker()) {
                    kind = SEG_RESYNC;
                    nextpos = r.idx;
                    break;
                }
                check = true;
                if (nblk >= cap) {
                    kind = SEG_CUT;
                    break;
                }
                if (r.idx > im.scan_len + 8u) { // in the zeros past the end: no check fires any more
                    kind = SEG_EOD;
                    break;
                }
                const uint32_t comp = seq_comp(im, (m0 + nblk) % im.nb);
                int diff;
                if (seq_block<false>(r, s_lut, ts, im, true, comp, 0u, &diff, nrec, nullptr)) {
                    kind = SEG_ERR;
                    break;
                }
                dc[comp] += (uint32_t)diff;
                nblk++;
            }
        }
        uint32_t nxt = 0;
        if (kind == SEG_RESYNC) { // the candidate at nextpos: it is one, by the rule of k_seg_list
            const uint32_t *pos = P.seg_pos + im.seg_off;
            uint32_t lo = 0, hi = cnt - 1u;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (pos[mid] < nextpos) lo = mid + 1u;
                else hi = mid;
            }
            nxt = lo + 1u;
        }
        P.seg_next[im.seg_node + node] = nxt | (((m0 + nblk) % im.nb) << 25) | (kind << 29);
        P.seg_w[im.seg_node + node] = make_uint4(nblk, nrec, (dc[0] & 0xFFFFu) | (dc[1] << 16), dc[2] & 0xFFFFu);
    }
}

// The true chain of one image: grid = the images with DRI; per node on the chain seg_base = (blocks, records before it, DC
// predictors there), seg_chain = the nodes in order.  The chain only ever moves to LATER candidates, so it is followed tile by
// tile (512 candidates x nb labels; 256 candidates for nb > 6), and inside a tile by pointer doubling in LDS instead of a walk:
// link level k of a node = the node 2^k segments on (none: it leaves the tile or ends), nine levels built in nine sweeps of the
// workgroup; then the chain's d-th node is filled in from its (d - 2^k)-th, the highest level first; then one prefix sum over
// the chain nodes' weights.  (A single lane that followed the links took 0.1 us per segment: 1.6 ms for the 16 200 one-MCU
// intervals of a 1080p 4:2:2 scan, as long as decoding all its segments under all labels.)
#define SEG_LEVELS 9u
struct SegSums {
    uint32_t b, r, d0, d1, d2;
};
__device__ __forceinline__ SegSums seg_add(const SegSums &x, const SegSums &y) { return SegSums{x.b + y.b, x.r + y.r, x.d0 + y.d0, x.d1 + y.d1, x.d2 + y.d2}; }

__global__ __launch_bounds__(256) void k_seg_chain(HjdParams P)
{
    __shared__ uint16_t s_lnk[SEG_LEVELS][SEG_TILE];
    __shared__ uint16_t s_d[512];
    __shared__ SegSums s_ws[4], s_carry;
    __shared__ uint32_t s_j, s_m, s_stop, s_stopat, s_w[17], s_flag;
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t cnt = P.seg_cnt[q];
    if (cnt == SEG_NONE) { // uniform
        if (seg_last_workgroup(P, 1u, &s_flag)) seg_plan(P, 1u, s_w);
        return;
    }
    const int ii = (int)P.rst_list[q];
    const HjdImage &im = P.img[ii];
    const uint32_t nb = im.nb, levels = nb <= 6u ? SEG_LEVELS : SEG_LEVELS - 1u, tile = 1u << levels; // candidates per tile: tile * nb <= SEG_TILE
    const uint32_t *gnext = P.seg_next + im.seg_node;
    const uint4 *gw = P.seg_w + im.seg_node;
    uint4 *gbase = P.seg_base + im.seg_node;
    uint32_t *gchain = P.seg_chain + im.seg_node;
    uint32_t nchain = 0; // uniform
    bool failed = false; // uniform
    if (tid == 0) s_j = 0u, s_m = 0u, s_stop = tile * nb <= SEG_TILE ? 0u : 2u, s_carry = SegSums{0u, 0u, 0u, 0u, 0u};
    __syncthreads();
    failed = s_stop == 2u; // more labels than a tile holds: not met (nb <= 10), left to k_seq_decode
    for (;;) { // uniform: one tile per turn, entered at candidate s_j with label s_m
        const uint32_t t0 = s_j, m_in = s_m;
        if (s_stop) break;
        const uint32_t n0 = t0 * nb, nn = (min(t0 + tile, cnt + 1u) - t0) * nb;
        for (uint32_t k = tid; k < nn; k += 256u) { // level 0: the segment's own successor, if it lies in the tile
            const uint32_t nx = gnext[n0 + k], jn = nx & 0x1FFFFFFu;
            s_lnk[0][k] = ((nx >> 29) == SEG_RESYNC && jn < t0 + tile) ? (uint16_t)((jn - t0) * nb + ((nx >> 25) & 15u)) : (uint16_t)0xFFFFu;
        }
        for (uint32_t lv = 1; lv < levels; lv++) {
            __syncthreads();
            for (uint32_t k = tid; k < nn; k += 256u) {
                const uint32_t a = s_lnk[lv - 1u][k];
                s_lnk[lv][k] = a == 0xFFFFu ? (uint16_t)0xFFFFu : s_lnk[lv - 1u][a];
            }
        }
        if (tid == 0) s_d[0] = (uint16_t)m_in, s_stopat = 0xFFFFFFFFu;
        for (int lv = (int)levels - 1; lv >= 0; lv--) { // the chain's node number d + 2^lv from its node number d
            __syncthreads();
            const uint32_t d = tid << (lv + 1);
            if (d < tile) {
                const uint32_t a = s_d[d];
                s_d[d + (1u << lv)] = a == 0xFFFFu ? (uint16_t)0xFFFFu : s_lnk[lv][a];
            }
        }
        __syncthreads();
        // prefix sums over the chain's nodes in this tile, 256 at a time
        for (uint32_t c0 = 0; c0 < tile; c0 += 256u) { // uniform
            const uint32_t i = c0 + tid, a = s_d[i];
            const bool valid = a != 0xFFFFu;
            uint32_t nx = 0;
            SegSums w{0u, 0u, 0u, 0u, 0u};
            if (valid) {
                nx = gnext[n0 + a];
                const uint4 q4 = gw[n0 + a];
                w = SegSums{q4.x, q4.y, q4.z & 0xFFFFu, q4.z >> 16, q4.w};
            }
            SegSums inc = w;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                SegSums o{__shfl_up(inc.b, dd, 64), __shfl_up(inc.r, dd, 64), __shfl_up(inc.d0, dd, 64), __shfl_up(inc.d1, dd, 64), __shfl_up(inc.d2, dd, 64)};
                if ((int)lane >= dd) inc = seg_add(inc, o);
            }
            if (lane == 63u) s_ws[wv] = inc;
            __syncthreads();
            SegSums ex = s_carry; // everything before this chunk
            for (uint32_t k = 0; k < wv; k++) ex = seg_add(ex, s_ws[k]);
            const SegSums tot = seg_add(seg_add(seg_add(seg_add(s_carry, s_ws[0]), s_ws[1]), s_ws[2]), s_ws[3]);
            ex = seg_add(ex, SegSums{inc.b - w.b, inc.r - w.r, inc.d0 - w.d0, inc.d1 - w.d1, inc.d2 - w.d2});
            if (valid) {
                const uint32_t kind = nx >> 29;
                gbase[n0 + a] = make_uint4(ex.b, ex.r, (ex.d0 & 0xFFFFu) | (ex.d1 << 16), ex.d2 & 0xFFFFu);
                // the chain ends with the first node that does not end in a resync, or with which the image's blocks are complete
                if (kind != SEG_RESYNC || ex.b + w.b >= im.nblocks) atomicMin(&s_stopat, i);
            }
            __syncthreads();
            const uint32_t stopat = s_stopat; // uniform
            const uint32_t nvalid = __syncthreads_count(valid ? 1 : 0); // chain nodes of this chunk (a prefix of it)
            const uint32_t take = stopat != 0xFFFFFFFFu ? stopat + 1u - c0 : nvalid; // ... of which belong to the chain
            if (valid && tid < take) gchain[nchain + tid] = n0 + a;
            if (stopat != 0xFFFFFFFFu) {
                if (i == stopat) {
                    const uint32_t kind = nx >> 29; // 2: a cut-off segment on the true chain before the image is complete: left to k_seq_decode
                    s_stop = ((kind == SEG_CUT && ex.b + w.b < im.nblocks) || kind == SEG_DEAD) ? 2u : 1u;
                }
                nchain += take;
                __syncthreads();
                break;
            }
            nchain += nvalid;
            if (tid == 0) s_carry = tot;
            if (valid && (tid + 1u == nvalid)) { // the last node so far: where the chain goes on if the tile ends here
                s_j = nx & 0x1FFFFFFu;
                s_m = (nx >> 25) & 15u;
            }
            __syncthreads();
            if (nvalid < 256u) break; // the chain has left the tile
        }
        __syncthreads();
        if (s_stop == 2u) failed = true;
    }
    if (tid == 0) {
        P.seg_nchain[q] = failed ? 0u : nchain;
        if (!failed) {
            HjdImageState &st = P.st[ii];
            st.flags &= HJD_FLAG_BAD_MARKER; // as k_seq_decode: the write pass adds what the true decode runs into
            st.total_blocks = im.nblocks;
            st.emulated = 2u;
        }
    }
    if (seg_last_workgroup(P, 1u, &s_flag)) seg_plan(P, 1u, s_w);
}

// the chain's nodes once more, writing: a persistent grid over the work items of seg_plan(1)
__global__ __launch_bounds__(SEG_WG) void k_seg_write(HjdParams P)
{
    __shared__ SeqLut s_lut[HJD_SLOTS];
    const uint32_t *plan = P.seg_plan + (P.nrst + 1u);
    const uint32_t total = plan[P.nrst];
    uint32_t have = 0xFFFFFFFFu;
    for (uint32_t item = blockIdx.x; item < total; item += gridDim.x) { // uniform
        const uint32_t q = seg_find(plan, P.nrst, item);
        const int ii = (int)P.rst_list[q];
        const HjdImage &im = P.img[ii];
        const HjdTableSet &ts = P.tabs[im.tabset];
        if (have != im.tabset) {
            __syncthreads();
            seq_load_lut(s_lut, ts);
            have = im.tabset;
            __syncthreads();
        }
        const uint32_t k = (item - plan[q]) * SEG_WG + threadIdx.x;
        if (k >= P.seg_nchain[q]) continue;
        const uint32_t node = P.seg_chain[im.seg_node + k];
        const uint4 base = P.seg_base[im.seg_node + node];
        const uint32_t j = node / im.nb, m0 = node % im.nb, b0 = base.x;
        SeqReader r;
        seg_start(r, P, im, j);
        uint32_t *rec = P.rec + im.rec_off;
        uint2 *blk = P.blk + im.blk_off;
        int16_t *dca = P.dca + im.blk_off;
        int prev[3] = {(int)(int16_t)(base.z & 0xFFFFu), (int)(int16_t)(base.z >> 16), (int)(int16_t)(base.w & 0xFFFFu)};
        uint32_t b = b0, nrec = base.y, flags = 0;
        bool check = j == 0u;
        while (b < im.nblocks) {
            if (check && r.on_marker()) break; // the next node of the chain starts here
            check = true;
            const uint32_t comp = seq_comp(im, (m0 + (b - b0)) % im.nb);
            const uint32_t first_rec = nrec;
            int diff;
            flags = seq_block<true>(r, s_lut, ts, im, true, comp, b, &diff, nrec, rec);
            if (flags) break;
            prev[comp] = (int)(int16_t)(diff + prev[comp]);
            blk[b] = make_uint2(first_rec, 0u);
            dca[b] = (int16_t)prev[comp];
            b++;
        }
        if (flags) { // the true decode fails here, before its last block: k_seq_decode's verdict
            atomicOr(&P.st[ii].flags, flags);
            P.st[ii].total_blocks = 0u;
        } else if (b == im.nblocks) blk[im.nblocks] = make_uint2(min(nrec, im.rec_cap), 0u);
    }
}

int hjd_launch_seq_decode(const HjdParams &P, uint32_t first, uint32_t n, hipStream_t s, int after_seg)
{
    if (!n) return 0;
    hipLaunchKernelGGL(k_seq_decode, dim3(n), dim3(64), 0, s, P, first, (uint32_t)after_seg);
    hipError_t e_ = hipGetLastError();
    return e_ == hipSuccess ? 0 : (int)e_;
}

// The segment-parallel emulation of every flagged image of rst_list that has node space (HjdImage::seg_cap); four launches, all
// of them gated on the device: for clean restart streams the workgroups of the first and the third return at once and the persistent grids find no work.
int hjd_launch_seg_emulation(const HjdParams &P, hipStream_t s)
{
    if (!P.nrst) return 0;
    const uint32_t grid = 1536u; // as many workgroups as the chip holds at once (six per CU: the tables' LDS)
    hipLaunchKernelGGL(k_seg_list, dim3(P.nrst), dim3(1024), 0, s, P);
    hipLaunchKernelGGL(k_seg_decode, dim3(grid), dim3(SEG_WG), 0, s, P);
    hipLaunchKernelGGL(k_seg_chain, dim3(P.nrst), dim3(256), 0, s, P);
    hipLaunchKernelGGL(k_seg_write, dim3(grid), dim3(SEG_WG), 0, s, P);
    hipError_t e_ = hipGetLastError();
    return e_ == hipSuccess ? 0 : (int)e_;
}
