// hjd_entropy.hip — Huffman VLC decode of the scan on gfx950 (MI355X, CDNA4).
//   k_sync    ProcessHuffmanBlock, speculative: find the decoder state at every subsequence start
//   k_repair  the same machinery re-run only where a workgroup's warm-up was too short
//   k_write   ProcessHuffmanBlock, final pass: coefficients out            (loadjpg.cpp:499-863)
//
// The scan of a baseline JPEG is one bit-serial string.  It is cut into fixed-size SUBSEQUENCES
// (S bits).  Huffman streams self-synchronise: a decoder started at a wrong position/state soon
// falls into step with the true decode.  Every subsequence is decoded from a guessed start state,
// then again from the end state of its predecessor, until start == predecessor's end everywhere in
// the workgroup.  Workgroups overlap by a few warm-up subsequences, so the first owned subsequence of
// every workgroup has (with overwhelming probability) the true state; k_write re-checks
// start == predecessor's end for EVERY subsequence across workgroup borders, and the host
// re-synchronises if that induction step ever fails.  Outputs are written only by k_write, so
// speculation never reaches them.
//
// Rounds are workgroup-synchronous and COMPACTED: after the two dense rounds (from the guess, from the
// predecessor) only a shrinking minority of subsequences still changes (measured on the q85 bench
// content: 30 %, 9 %, 3 % ...).  Each round the subsequences whose start state changed are gathered
// (__ballot + prefix over the waves) and handed to the first lanes of the workgroup, so a round with 40
// changed subsequences costs one wave's issue slots instead of eight waves each dragging a few live
// lanes through the whole loop (round 1's kernel spent 5.9 wave-passes per wave for 2.8 lane-passes of
// work).  Waves without work wait at the barrier and issue nothing.
//
// Stream layout (written by k_unstuff_write): TRANSPOSED in groups of 64 subsequences — word k of
// subsequence i lives at word ((i/64)*(wps+2) + k)*64 + i%64, and every row is extended by the
// first two words of the next row — so that 64 consecutive subsequences are staged with fully coalesced
// loads.  A lane stages only a WINDOW of its row (phase_words + 2 words) into LDS ([k*64 + lane],
// conflict-free) and decodes it before staging the next one: the decode loop touches LDS only (its
// stores are never waited for — loads and stores share vmcnt on gfx9-class hardware), and the LDS a
// workgroup holds is small enough for 6 waves per SIMD (round 1 parked whole rows: 45 KB per 256
// threads, 1.4 resident waves per SIMD on average).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <type_traits>

#include "hjd_device.h"
#include "hjd_kernels.h"
#include "hjd_kutil.h"

__constant__ uint8_t c_zz_to_slot[64] = HJD_ZZ_TO_SLOT;

struct EntropyLds {
    const uint16_t *lut; // [nlut][HJD_LUT_SIZE]  (synchronisation form: [nlut][HJD_LUT_SIZE][2], see HjdTableSet::lut_s)
    const uint16_t *sec; // [nsec][HJD_SEC_SIZE]
    const uint2 *mt;     // [nb] block-in-MCU m -> (byte offsets DC | AC << 16 of its tables, next m): the block-end step is
                         // taken by some lane of a wave in nearly every iteration, so it is two LDS reads instead of nine VALU
    uint32_t mt4_addr, sec_addr, slot_addr; // k_write only: LDS byte addresses of the hand-written loop's block table
                                            // (fill_block_table), of the second-level tables and of the slot table
};

// fills EntropyLds::mt (first nb lanes; the caller synchronises)
__device__ __forceinline__ void fill_mcu_table(uint2 *mt, uint32_t nb, uint32_t nby, uint32_t tabY, uint32_t tabCb, uint32_t tabCr)
{
    const uint32_t m = threadIdx.x;
    if (m < nb) mt[m] = make_uint2(m < nby ? tabY : (m == nby ? tabCb : tabCr), m + 1u == nb ? 0u : m + 1u);
}

// address of word 0 of subsequence (row) i in the transposed stream; successive words are 64 words apart
__device__ __forceinline__ const uint32_t *row_base(const uint32_t *w, uint32_t i, uint32_t lw)
{
    return w + ((size_t)(i >> 6) * ((1u << lw) + 2u)) * 64u + (i & 63u);
}

template <bool SYNC_FORM>
__device__ __forceinline__ void load_tables(const HjdTableSet *ts, void *s_lut, void *s_sec)
{
    const uint32_t nl = ts->nlut * ((SYNC_FORM ? HJD_LUT_SIZE_S * 4 : HJD_LUT_SIZE * 2) / 16);
    const uint32_t ns = SYNC_FORM ? ts->nsec_s * (HJD_SEC_SIZE_S * 2 / 16) : ts->nsec * (HJD_SEC_SIZE * 2 / 16);
    const uint4 *src = reinterpret_cast<const uint4 *>(SYNC_FORM ? &ts->lut_s[0][0][0] : &ts->lut[0][0]);
    uint4 *dst = reinterpret_cast<uint4 *>(s_lut);
    for (uint32_t k = threadIdx.x; k < nl; k += blockDim.x) dst[k] = src[k];
    src = reinterpret_cast<const uint4 *>(SYNC_FORM ? &ts->sec_s[0][0] : &ts->sec[0][0]);
    dst = reinterpret_cast<uint4 *>(s_sec);
    for (uint32_t k = threadIdx.x; k < ns; k += blockDim.x) dst[k] = src[k];
}

// (DC | AC << 16) byte offsets of a component's primary tables, from the packed slot map; entry_bytes = 2, or 4 for
// the synchronisation form with its two entries per index
__device__ __forceinline__ uint32_t table_pair(uint32_t slot_pack, uint32_t comp, uint32_t entry_bytes)
{
    const uint32_t tbytes = entry_bytes == 4u ? HJD_LUT_SIZE_S * 4u : HJD_LUT_SIZE * 2u; // bytes of one primary table
    const uint32_t dc = ((slot_pack >> (comp * 8u)) & 7u) * tbytes;
    const uint32_t ac = ((slot_pack >> (comp * 8u + 4u)) & 7u) * tbytes;
    return dc | (ac << 16);
}

// ---- bit reader --------------------------------------------------------------------------------
// window() = the next 32 bits; advance(n) consumes n <= 31 bits.  State: words hi, lo and sh = number
// of unconsumed bits of hi (0..31); window = alignbit(hi, lo, sh).  `rowp` points at word 0 of the
// staged window of the lane's row (64 words between successive words).
struct ReaderLds {
    uint32_t hi, lo, nxt, sh;
    const uint32_t *nx;
    uint32_t stride; // words between successive words of the row in LDS: 64 in the staging windows of k_sync, rows + 1 in k_write
    __device__ __forceinline__ void init(const uint32_t *rowp, uint32_t rel, uint32_t stride_words)
    {
        stride = stride_words;
        const uint32_t k0 = rel >> 5, o0 = rel & 31u;
        const uint32_t *sp = rowp + k0 * stride;
        const uint32_t a = sp[0], b = sp[stride];
        hi = a;
        lo = o0 ? b : a;
        nx = sp + (o0 ? 2u * stride : stride);
        sh = (32u - o0) & 31u;
        nxt = 0;
    }
    __device__ __forceinline__ void prefetch() { nxt = *nx; } // fetched alongside the table lookup
    __device__ __forceinline__ uint32_t window() const { return __builtin_amdgcn_alignbit(hi, lo, sh); }
    __device__ __forceinline__ void advance(uint32_t tot)
    {
        const int sh2 = (int)sh - (int)tot;
        const bool roll = sh2 < 0;
        hi = roll ? lo : hi;
        lo = roll ? nxt : lo;
        nx += roll ? stride : 0u;
        sh = (uint32_t)sh2 & 31u;
    }
};

// RESTART (images with DRI): the reference resynchronises whenever its byte cursor sits on a marker
// at a block start (loadjpg.cpp:535-550) — i.e. when a block ends inside the last data byte before
// a marker — and between a DC code and its extra bits (loadjpg.cpp:631-640); DC predictors are NOT
// reset.  Inside the clean-resync domain that equals: at the end of the last
// block of an interval jump to the marker position; everything else the reference would do with a
// marker (firing early on tiny trailing blocks, swallowing markers as data) is flagged DIRTY.
// Speculative decodes are simply pulled back onto the next interval start when they run over a
// marker, which also ends every synchronisation chain there.
struct RestartCtx {
    const uint32_t *mpos; // marker bit positions of the image
    uint32_t nmark;
    uint32_t mk;          // index of the next marker (first position > p)
    uint32_t ri_blocks;   // blocks per interval, 0 = no intervals expected
    uint32_t left;        // WRITE: blocks left in the current interval
    uint32_t conf;        // conformant mode (hjd_options.conformant): the standard's interval rule instead of the reference's
    uint32_t mnext, mnext2; // WRITE: positions of markers mk and mk + 1 (all ones: none), kept a marker ahead so that a jump does not wait for a load
};

// WRITE: moves rc.mk to the first marker past p and leaves its position and the next one's in rc.mnext / rc.mnext2.  Four entries per
// round trip (a lane starts from its subsequence's first marker — k_sync's submk — and a subsequence of a stream with a marker per
// MCU holds a dozen: one dependent load each was 20 us of a lane's 30).
__device__ __forceinline__ void marker_seek(RestartCtx &rc, uint32_t p)
{
    for (;;) {
        uint32_t v[5];
#pragma unroll
        for (uint32_t q = 0; q < 5u; q++) v[q] = rc.mk + q < rc.nmark ? rc.mpos[rc.mk + q] : 0xFFFFFFFFu;
        const uint32_t cnt = (v[0] <= p ? 1u : 0u) + (v[1] <= p ? 1u : 0u) + (v[2] <= p ? 1u : 0u) + (v[3] <= p ? 1u : 0u);
        if (cnt < 4u) {
            rc.mk += cnt;
            rc.mnext = cnt == 0u ? v[0] : (cnt == 1u ? v[1] : (cnt == 2u ? v[2] : v[3]));
            rc.mnext2 = cnt == 0u ? v[1] : (cnt == 1u ? v[2] : (cnt == 2u ? v[3] : v[4]));
            return;
        }
        rc.mk += 4u;
    }
}

// WRITE: where the outputs go.  A lane's records wait for their store in its column of an LDS ring (word k of the burst at
// ring[k * HJD_WRITE_WG]): every lane's run of records starts at a multiple of HJD_REC_RUN, so a full burst is two aligned
// global_store_dwordx4 back to back, one 32-byte sector.  (Until round 3 the burst was a shift register of eight VGPRs: fourteen
// v_mov per loop iteration around the record branch, whether the iteration produced a record or not.)
// 16 bytes of records.  (Streamed with non-temporal stores — tried this round, to keep the record array out of the L2 — k_write
// took 0.28 ms instead of 0.15: a lane's 16-byte pieces then leave one by one.)
typedef uint32_t hjd_rec4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_records(uint32_t *dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    *reinterpret_cast<hjd_rec4 *>(dst) = (hjd_rec4){a, b, c, d};
}

struct WriteCtx {
    uint32_t *ring;
    uint32_t *rec;
    uint32_t rec_cap;
    uint2 *blk;
    uint32_t nblocks;
    const uint8_t *slot_of; // LDS: zig-zag index -> slot u*8+v
    uint32_t bfirst;        // first record of the block in progress, or ~0 when the block began in an earlier span
    uint32_t bdc;
    uint32_t tag;           // block bits of a record: (block & 127) << 22
    uint32_t *ent;          // LDS: the workgroup's block entries in the making (HJD_ENT_CAP words, all ones = not written here)
    uint32_t b0, r0;        // the workgroup's first block / first record
};

// (defined below, behind the LDS helpers it uses)
template <uint32_t WGSZ, uint32_t ENTCAP>
__device__ __forceinline__ void write_window_plain(uint32_t mt4_addr, uint32_t sec_addr, uint32_t slot_addr, const uint32_t *__restrict__ rowp,
                                                   uint32_t stride, uint32_t row_bit0, uint32_t &p, uint32_t &z, uint32_t &m, uint32_t end,
                                                   uint32_t &nblk, uint32_t &nrec, WriteCtx &W, uint32_t &flags);

// The entry of block b = (first record, DC difference), made by the lane that decoded the block's DC symbol (bfirst != ~0), when
// the block ends or the lane's span does.  A workgroup's blocks are consecutive, so the entries wait in LDS — 4 bytes: first
// record relative to the workgroup's first | DC difference << 16 — and leave together at the end, whole lines of them (k_write).
// (Rounds 2-3 stored 16 bytes per block, with the mask of the slots the block's records fill, plus the DC difference once more
// into a 2-byte array for the DC prediction kernels; every such lone store cost a 32-byte write: k_write wrote 1.8 times its
// useful bytes.  The back end builds the masks from the records themselves — it touches every record anyway.)
template <uint32_t ENTCAP>
__device__ __forceinline__ void write_block_entry(WriteCtx &W, uint32_t b)
{
    if (W.bfirst != 0xFFFFFFFFu) {
        const uint32_t idx = b - W.b0, rel = W.bfirst - W.r0;
        if (idx < ENTCAP && rel < 0xFFFFu) W.ent[idx] = rel | (W.bdc << 16);
        else W.blk[b] = make_uint2(W.bfirst, W.bdc & 0xFFFFu); // more blocks or records than the staging holds: straight out
    }
    W.bfirst = 0xFFFFFFFFu;
}

// Synchronisation form of the decode loop: symbols from state (p, z, m) until p >= end; nblk counts blocks completed, nrec the
// AC records they hold; no values are extracted.  `rowp` points at the staged window of the lane's row, `row_bit0` is the
// absolute bit position of the window's first bit.  tabY/tabCb/tabCr: byte offsets of the component's (DC | AC << 16)
// primary tables.
// RESTART = true is the loop that knows the reference's marker rules; it runs only while a marker is within reach of them
// (p + 40 > the next marker's position: a step is at most 31 bits and the rules look 8 bits ahead) — sync_span hands the long
// stretches between markers to the plain loop.
template <bool RESTART>
__device__ __forceinline__ void sync_core(const EntropyLds &L, const uint32_t *__restrict__ rowp, uint32_t row_bit0, uint32_t tabY,
                                          uint32_t tabCb, uint32_t tabCr, uint32_t nby, uint32_t nb, uint32_t &p, uint32_t &z, uint32_t &m,
                                          uint32_t end, uint32_t pair_lim, uint32_t &nblk, uint32_t &nrec, RestartCtx &rc)
{
    ReaderLds rd;
    rd.init(rowp, p - row_bit0, 64u);
    uint32_t mnext = 0xFFFFFFFFu;
    if (RESTART) {
        while (rc.mk < rc.nmark && rc.mpos[rc.mk] <= p) rc.mk++;
        mnext = rc.mk < rc.nmark ? rc.mpos[rc.mk] : 0xFFFFFFFFu;
    }
    uint32_t pair = m < nby ? tabY : (m == nby ? tabCb : tabCr);
    uint32_t mnx = (m + 1u == nb) ? 0u : m + 1u;
    uint2 nxt = L.mt[mnx]; // the NEXT block's table pair and successor, fetched a block ahead: no LDS round trip in the block-end step
    uint32_t tb = z ? pair >> 16 : pair & 0xFFFFu;
    const char *lutb = reinterpret_cast<const char *>(L.lut);

    while (p < end && (!RESTART || p + 40u > mnext)) {
        rd.prefetch();
        const uint32_t win = rd.window();
        // [0] this symbol, [1] this symbol and the next as one step (HjdTableSet::lut_s).  The pair is taken while the first
        // symbol cannot end the block (an AC advance is at most 16) and the second still starts inside the window (a step
        // is at most 31 bits): the states passed on are then exactly those of the symbol-by-symbol walk of the final pass.
        const uint32_t ep = *reinterpret_cast<const uint32_t *>(lutb + tb + ((win >> (32 - HJD_LUT_BITS_S)) << 2));
        const uint32_t e1 = ep & 0xFFFFu, e2 = ep >> 16;
        // restart streams: only while the next marker is more than a step (31 bits) plus the 8 bits of the rules' look-ahead
        // away, so that no marker rule could have fired at the boundary the pair steps over
        uint32_t e = (z < 48u && p < pair_lim && (!RESTART || p + 40u < mnext)) ? e2 : e1;
        if ((e & 31u) == 0u) {
            // code longer than HJD_LUT_BITS_S: second-level table on the next HJD_SEC_BITS_S bits (IsInHuffmanCodes,
            // loadjpg.cpp:335-392, for the lengths the primary table cannot hold); 0 = no such code
            const uint32_t sub = e >> 5;
            e = sub ? L.sec[(sub - 1u) * HJD_SEC_SIZE_S + ((win >> 16) & (HJD_SEC_SIZE_S - 1u))] : 0u;
            if ((e & 31u) == 0u) e = 16u; // skip 16 bits: any progress will do, the final pass flags it
        }
        // e = (length + value bits) | advance << 5 | records << 14
        const uint32_t tot = e & 31u, adv = (e >> 5) & 127u;
        const bool isdc = z == 0u;
        nrec += e >> 14;
        z += adv;
        tb = isdc ? pair >> 16 : tb;
        const bool blk_end = z >= 64u;
        if (blk_end) { // EOB or last coefficient: next block (DecodeMCU order, loadjpg.cpp:945-997)
            z = 0u;
            nblk++;
            m = mnx;
            pair = nxt.x;
            mnx = nxt.y;
            nxt = L.mt[mnx];
            tb = pair & 0xFFFFu;
        }
        p += tot;
        rd.advance(tot);
        if (RESTART && rc.conf) {
            // Conformant mode (never the default): ITU T.81 B.2.4.1 / F.1.2.3 — an interval ends where nothing but
            // padding (1-bits up to the byte boundary) separates the end of a block from the marker.  Decidable
            // from the bits alone, so the speculative and the final pass apply the same rule.
            bool jump = p > mnext;
            if (blk_end && !jump) {
                const uint32_t d = mnext - p;
                if (d < 8u) jump = d == 0u || (rd.window() >> (32u - d)) == (1u << d) - 1u;
            }
            if (jump) {
                p = mnext;
                z = 0u;
                m = 0u;
                pair = tabY;
                mnx = L.mt[0].y;
                nxt = L.mt[mnx];
                tb = pair & 0xFFFFu;
                rc.mk++;
                mnext = rc.mk < rc.nmark ? rc.mpos[rc.mk] : 0xFFFFFFFFu;
                if (p >= end) break;
                rd.init(rowp, p - row_bit0, 64u);
            }
        } else if (RESTART) {
            if ((blk_end && p + 8u > mnext) || p > mnext) { // pulled back onto the next interval start
                p = mnext;
                z = 0u;
                m = 0u;
                pair = tabY;
                mnx = L.mt[0].y;
                nxt = L.mt[mnx];
                tb = pair & 0xFFFFu;
                rc.mk++;
                mnext = rc.mk < rc.nmark ? rc.mpos[rc.mk] : 0xFFFFFFFFu;
                if (p >= end) break; // the next window is staged by the caller
                rd.init(rowp, p - row_bit0, 64u);
            }
        }
    }
}

template <bool RESTART>
__device__ __forceinline__ void sync_span(const EntropyLds &L, const uint32_t *__restrict__ rowp, uint32_t row_bit0, uint32_t tabY,
                                          uint32_t tabCb, uint32_t tabCr, uint32_t nby, uint32_t nb, uint32_t &p, uint32_t &z, uint32_t &m,
                                          uint32_t end, uint32_t pair_lim, uint32_t &nblk, uint32_t &nrec, RestartCtx &rc)
{
    if (!RESTART) {
        sync_core<false>(L, rowp, row_bit0, tabY, tabCb, tabCr, nby, nb, p, z, m, end, pair_lim, nblk, nrec, rc);
        return;
    }
    // Restart streams (round 2 evaluated the marker rules at every step of every lane: 96 against 134 Gpixel/s): the plain loop
    // up to 39 bits before the next marker, the loop with the rules around it.
    while (p < end) {
        while (rc.mk < rc.nmark && rc.mpos[rc.mk] <= p) rc.mk++;
        const uint32_t mnext = rc.mk < rc.nmark ? rc.mpos[rc.mk] : 0xFFFFFFFFu;
        if (p + 40u <= mnext) {
            const uint32_t fend = min(end, mnext - 39u);
            sync_core<false>(L, rowp, row_bit0, tabY, tabCb, tabCr, nby, nb, p, z, m, fend, min(pair_lim, fend - 1u), nblk, nrec, rc); // pairs while p + 40 < mnext
            if (p >= end) break;
        }
        sync_core<true>(L, rowp, row_bit0, tabY, tabCb, tabCr, nby, nb, p, z, m, end, pair_lim, nblk, nrec, rc);
    }
}

// The final pass over one staging window: decode symbols from state (p, z, m) until p >= end, one symbol per iteration, and
// write the coefficients out (output format: WriteCtx, hjd_device.h).  nblk = absolute block index, nrec = index of the next
// record.  Everything a DC symbol does is predicated (selects) instead of branched: some lane of a wave is at a DC symbol, at
// an end of block and at an AC coefficient in nearly every iteration, so every branch region is executed by every wave anyway
// (round 2's loop: ~90 vector instructions per iteration, a third of them moves and mask juggling around the regions).
template <bool RESTART, uint32_t WGSZ, uint32_t ENTCAP, bool WHOLE = false, bool CONF = false>
__device__ __forceinline__ void write_core(const EntropyLds &L, const uint32_t *__restrict__ rowp, uint32_t stride, uint32_t row_bit0, uint32_t tabY,
                                           uint32_t tabCb, uint32_t tabCr, uint32_t nby, uint32_t nb, uint32_t &p, uint32_t &z, uint32_t &m,
                                           uint32_t end, uint32_t &nblk, uint32_t &nrec, WriteCtx &W, uint32_t &flags, RestartCtx &rc)
{
    ReaderLds rd;
    rd.init(rowp, p - row_bit0, stride);
    uint32_t mnext = 0xFFFFFFFFu;
    if (RESTART) {
        if (!WHOLE) marker_seek(rc, p); // (WHOLE: write_span has just done it)
        mnext = rc.mnext;
    }
    uint32_t pair = m < nby ? tabY : (m == nby ? tabCb : tabCr);
    uint32_t mnx = (m + 1u == nb) ? 0u : m + 1u;
    uint2 nxt = L.mt[mnx]; // the NEXT block's table pair and successor, fetched a block ahead
    uint32_t tb = z ? pair >> 16 : pair & 0xFFFFu;
    const char *lutb = reinterpret_cast<const char *>(L.lut);

    while (p < end && (!RESTART || WHOLE || p + 40u > mnext)) { // WHOLE: the loop with the rules over the whole span (write_span)
        rd.prefetch();
        const uint32_t win = rd.window();
        uint32_t e = *reinterpret_cast<const uint16_t *>(lutb + tb + ((win >> (32 - HJD_LUT_BITS)) << 1));
        if ((e & 31u) == 0u) {
            // code longer than HJD_LUT_BITS: second-level table on the next HJD_SEC_BITS bits (IsInHuffmanCodes,
            // loadjpg.cpp:335-392, for the lengths the primary table cannot hold); 0 = no such code
            const uint32_t sub = e >> 5;
            e = sub ? L.sec[(sub - 1u) * HJD_SEC_SIZE + ((win >> 16) & (HJD_SEC_SIZE - 1u))] : 0u;
            if ((e & 31u) == 0u) {
                flags |= HJD_FLAG_BAD_CODE;
                e = 16u; // skip 16 bits: any progress will do, the result is flagged
            }
        }
        // e = code length + value bits | value bits << 5 | advance << 9
        const uint32_t s = (e >> 5) & 15u, adv = e >> 9, len = (e & 31u) - s;
        // value bits + DetermineSign (loadjpg.cpp:396-409): v < 2^(s-1), i.e. leading value bit 0, means v - (2^s - 1)
        const uint32_t t = win << len;
        const uint32_t vb = (t >> 1) >> (31u - s);
        const int val = (int)vb - (int)(~(uint32_t)((int)t >> 31) & ((1u << s) - 1u));
        const bool isdc = z == 0u;
        // DC symbol (loadjpg.cpp:562-667): the block's entry begins here
        W.bfirst = isdc ? nrec : W.bfirst;
        W.bdc = isdc ? (uint32_t)(int)(int16_t)val : W.bdc;
        // (W.tag = the block bits of the block in progress: set by the caller and at every block's end — computing it here on every
        // DC symbol was three instructions in every iteration)
        // the reference looks for a marker again between the DC code and its extra bits
        if (RESTART && !CONF && isdc && s != 0u && p + len + 8u > mnext) flags |= HJD_FLAG_DIRTY;
        const uint32_t zw = z + adv - 1u;
        if (!isdc && s != 0u) { // AC run/size: loadjpg.cpp:776-808
            const uint32_t slot = W.slot_of[zw & 63u];
            const uint32_t w = ((uint32_t)val & 0xFFFFu) | (slot << 16) | W.tag;
            // zw > 63: the reference writes DCT[nr] out of bounds here (:780-806): flagged — zw <= 63 + 15, so bit 6 says it; the
            // record (of slot zw & 63) stays where it is: the image's output is not looked at
            static_assert(HJD_FLAG_OVERRUN == 2u, "bit 6 of zw moved to bit 1");
            flags |= (zw >> 5) & HJD_FLAG_OVERRUN;
            static_assert(HJD_REC_RUN % 4 == 0 && (HJD_REC_RUN & (HJD_REC_RUN - 1)) == 0, "16-byte pieces");
            W.ring[(nrec & (HJD_REC_RUN - 1u)) * WGSZ] = w;
            if ((nrec & (HJD_REC_RUN - 1u)) == HJD_REC_RUN - 1u) { // a burst is complete: whole, aligned sectors
                if (nrec < W.rec_cap) {
                    uint32_t *dst = W.rec + (nrec - (HJD_REC_RUN - 1u));
#pragma unroll
                    for (uint32_t q = 0; q < HJD_REC_RUN / 4u; q++)
                        store_records(dst + 4u * q, W.ring[(4u * q) * WGSZ], W.ring[(4u * q + 1u) * WGSZ],
                                      W.ring[(4u * q + 2u) * WGSZ], W.ring[(4u * q + 3u) * WGSZ]);
                } else {
                    flags |= HJD_FLAG_OVERRUN;
                }
            }
            nrec++;
        }
        z += adv;
        tb = isdc ? pair >> 16 : tb;
        const bool blk_end = z >= 64u;
        const uint32_t tot = len + s;
        p += tot;
        rd.advance(tot);
        if (blk_end) { // EOB or last coefficient: next block (DecodeMCU order, loadjpg.cpp:945-997)
            z = 0u;
            write_block_entry<ENTCAP>(W, nblk);
            nblk++;
            W.tag = (nblk & 127u) << 22;
            if (nblk == W.nblocks) {
                W.blk[nblk] = make_uint2(nrec, 0u); // end of the last block's records
                end = p;                                    // the image is complete: what follows is not looked at
            }
            m = mnx;
            pair = nxt.x;
            mnx = nxt.y;
            nxt = L.mt[mnx];
            tb = pair & 0xFFFFu;
        }
        if (RESTART && CONF) { // (a template parameter: as a run-time branch the two rule sets made the loop a third longer)
            // Conformant mode (never the default): ITU T.81 B.2.4.1 / F.1.2.3 — an interval ends where nothing but
            // padding (1-bits up to the byte boundary) separates the end of a block from the marker.
            bool jump = p > mnext;
            if (blk_end && !jump) {
                const uint32_t d = mnext - p;
                if (d < 8u) jump = d == 0u || (rd.window() >> (32u - d)) == (1u << d) - 1u;
            }
            if (jump) {
                p = mnext;
                z = 0u;
                m = 0u;
                pair = tabY;
                mnx = L.mt[0].y;
                nxt = L.mt[mnx];
                tb = pair & 0xFFFFu;
                rc.mk++;
                mnext = rc.mnext2;
                rc.mnext2 = rc.mk + 1u < rc.nmark ? rc.mpos[rc.mk + 1u] : 0xFFFFFFFFu; // needed at the next jump only
                if (p >= end) break;
                rd.init(rowp, p - row_bit0, stride);
            }
        } else if (RESTART) {
            if (blk_end) rc.left--;
            if ((blk_end && p + 8u > mnext) || p > mnext) {
                if (nblk < W.nblocks) {
                    // clean only if a whole interval of whole MCUs ends exactly here.  ri_blocks = 0: the interval is not known
                    // (HjdImage::ri_blocks) — the reference's check fires at this block start whatever the count
                    // (loadjpg.cpp:535-550 tests the two bytes under its cursor and nothing else) and, the MCU being complete,
                    // goes on exactly as this loop does: marker skipped, reservoir dropped, first block of an MCU next
                    const bool clean = blk_end && p <= mnext && m == 0u && (rc.ri_blocks == 0u || rc.left == 0u);
                    if (!clean) flags |= HJD_FLAG_DIRTY;
                }
                p = mnext;
                z = 0u;
                m = 0u;
                pair = tabY;
                mnx = L.mt[0].y;
                nxt = L.mt[mnx];
                tb = pair & 0xFFFFu;
                rc.left = rc.ri_blocks;
                rc.mk++;
                mnext = rc.mnext2;
                rc.mnext2 = rc.mk + 1u < rc.nmark ? rc.mpos[rc.mk + 1u] : 0xFFFFFFFFu; // needed at the next jump only
                if (p >= end) break; // the next window is staged by the caller
                rd.init(rowp, p - row_bit0, stride);
            } else if (blk_end && rc.ri_blocks != 0u && rc.left == 0u && nblk < W.nblocks) {
                flags |= HJD_FLAG_DIRTY; // an interval ends here but the marker is not where the reference looks
                rc.left = rc.ri_blocks;
            }
        }
    }
}

template <bool RESTART, uint32_t WGSZ, uint32_t ENTCAP>
__device__ __forceinline__ void write_span(const EntropyLds &L, const uint32_t *__restrict__ rowp, uint32_t stride, uint32_t row_bit0, uint32_t tabY,
                                           uint32_t tabCb, uint32_t tabCr, uint32_t nby, uint32_t nb, uint32_t &p, uint32_t &z, uint32_t &m,
                                           uint32_t end, uint32_t &nblk, uint32_t &nrec, WriteCtx &W, uint32_t &flags, RestartCtx &rc)
{
    if (!RESTART) {
        write_core<false, WGSZ, ENTCAP>(L, rowp, stride, row_bit0, tabY, tabCb, tabCr, nby, nb, p, z, m, end, nblk, nrec, W, flags, rc);
        return;
    }
    // A wave none of whose lanes meets a marker inside its span (nor comes within reach of one: a step is at most 31 bits, the rules
    // look 8 bits ahead) runs the plain loop; any other wave runs the loop that knows the reference's rules (loadjpg.cpp:535-550,
    // :631-640) over the whole span, ALL its lanes together.  (Rounds 2-4: the plain loop up to 39 bits before a lane's next marker,
    // the loop with the rules around it, the plain loop again — the few lanes with a marker walked the rest of their span while the
    // others of their wave waited: k_write 0.196 ms against 0.148 without DRI for 64 x 1080p with one marker per MCU row; round 2's
    // first form ran the rules in every wave of every restart stream.)
    marker_seek(rc, p);
    const uint32_t mnext = rc.mnext;
    const bool reach = p < end && nblk < W.nblocks && mnext < end + 40u;
    if (!__any(reach)) {
        // no marker in sight: the count of the blocks left in the interval is brought up to date afterwards, and an interval that ended
        // inside the span — no marker where the block count says one should be — is the DIRTY verdict the loop with the rules gives
        // (loadjpg.cpp:535-550 never fires there, the reference decodes on)
        const uint32_t before = nblk;
        if (p < end) write_window_plain<WGSZ, ENTCAP>(L.mt4_addr, L.sec_addr, L.slot_addr, rowp, stride, row_bit0, p, z, m, end, nblk, nrec, W, flags);
        const uint32_t done = nblk - before;
        if (!rc.conf && rc.ri_blocks != 0u) {
            if (done >= rc.left && before + rc.left < W.nblocks) {
                flags |= HJD_FLAG_DIRTY;
                rc.left = rc.ri_blocks - (done - rc.left) % rc.ri_blocks;
            } else {
                rc.left -= done;
            }
        }
        return;
    }
    if (nblk >= W.nblocks) return;
    if (rc.conf) write_core<true, WGSZ, ENTCAP, true, true>(L, rowp, stride, row_bit0, tabY, tabCb, tabCr, nby, nb, p, z, m, end, nblk, nrec, W, flags, rc);
    else write_core<true, WGSZ, ENTCAP, true, false>(L, rowp, stride, row_bit0, tabY, tabCb, tabCr, nby, nb, p, z, m, end, nblk, nrec, W, flags, rc);
}

// ---- stage 1: synchronisation ----------------------------------------------------------------
// A synchronisation workgroup holds HJD_SYNC_WG SLOTS = subsequences, HJD_SYNC_C per lane (slot (wv * C + c) * 64 + lane is chain c
// of that lane, so that the 64 lanes of a wave hold 64 consecutive subsequences per chain: coalesced staging out of the
// transposed stream).  What a lane does with a staging window is ONE loop (sync_window) in which nothing branches on a lane's data:
// a chain that has reached the end of its window adds an entry of zero, the block-end work (next block's tables) is three
// selects on values fetched unconditionally from a small LDS table, the table pair (this symbol alone / this symbol and the
// next) is a select.  EXEC stays full, the loop is left when no lane of the wave has bits left, and the only branch inside is
// the wave-uniform one to the second-level table (a code longer than HJD_LUT_BITS: 0.2 % of the symbols).
//
// What this loop costs is its instruction count: a wave issues ONE instruction of any kind every 4-5 cycles whatever the other
// waves of its SIMD do (measured here: round 3's loop, ~80 instructions per step with its divergent regions, 500 cycles per step;
// the compiler's version of the loop below 58 and 385; the hand-written one 42 and 330 with two waves per SIMD).  Several
// chains per lane (HJD_SYNC_C = 2, built and measured this round: the look-ups of one chain in flight while the other's
// arithmetic issues) shorten nothing: a workgroup then has half the waves, each with twice the instructions, and a SIMD with
// one resident wave issues at half the rate of one with two — k_sync 0.47 ms against 0.32 for 16 images.  Waves, not chains
// per wave, are what fills a SIMD; HJD_SYNC_C stays 1.
//
// Restart streams (images with DRI) keep the loop that knows the reference's marker rules (sync_span<true>).
//
// MODE 0: workgroup w of an image owns subsequences [w*Q, (w+1)*Q), Q = HJD_SYNC_WG - K, and additionally
//         decodes the K subsequences before them as warm-up (their records belong to workgroup w-1).
// MODE 1: repair / host re-synchronisation: HJD_SYNC_WG owned subsequences per workgroup, records reloaded,
//         slot 0 takes its start from the previous workgroup's last record.  Returns at once (before the
//         tables are even loaded) when every start state already equals its predecessor's end.
// MODE 2: MODE 1 for the last repair launch, which also leaves the block / record totals of the final-pass
//         workgroups it covers.

struct SyncChain {
    uint32_t p;        // bit position inside the staging window
    uint32_t z;        // zig-zag index
    uint32_t sh, hi, lo;
    uint32_t nxa;      // LDS byte address of the next stream word (prefetched every step)
    uint32_t dcb, acb; // LDS byte addresses of the current block's DC / AC table
    uint32_t mi;       // LDS byte address of the block table's entry for the NEXT block of the MCU
    uint32_t nblk;     // blocks completed in the window
    uint32_t nrec;     // AC records in the window
};

// Block table of the synchronisation kernels, one uint4 per block-in-MCU index m: x / y = LDS byte addresses of its DC / AC table,
// z = LDS byte address of the entry of m + 1 (cyclic), w = m - 1 (cyclic): what a chain whose `mi` points HERE is decoding.
__device__ __forceinline__ void fill_block_table(uint4 *mt4, uint32_t lds_addr_of_mt4, uint32_t nb, uint32_t nby, uint32_t lut_addr,
                                                 uint32_t tabY, uint32_t tabCb, uint32_t tabCr, uint32_t t)
{
    if (t < nb) {
        const uint32_t pair = t < nby ? tabY : (t == nby ? tabCb : tabCr); // byte offsets DC | AC << 16
        mt4[t] = make_uint4(lut_addr + (pair & 0xFFFFu), lut_addr + (pair >> 16), lds_addr_of_mt4 + (t + 1u == nb ? 0u : t + 1u) * 16u,
                            t == 0u ? nb - 1u : t - 1u);
    }
}

__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)p; // generic -> LDS: the low 32 bits are the LDS byte address
}
// loads by LDS byte address (ds_read_b16 / b32 / b128)
typedef uint32_t hjd_u32x4 __attribute__((ext_vector_type(4)));
template <class T>
__device__ __forceinline__ T lds_load(uint32_t a)
{
    return *(const __attribute__((address_space(3))) T *)(uintptr_t)a;
}
template <>
__device__ __forceinline__ uint4 lds_load<uint4>(uint32_t a)
{
    const hjd_u32x4 v = *(const __attribute__((address_space(3))) hjd_u32x4 *)(uintptr_t)a;
    return make_uint4(v.x, v.y, v.z, v.w);
}

// The plain form of write_core (no restart markers in sight), instruction by instruction (round 5), for the same reason as
// sync_window<1>: what this loop costs is its instruction count.  Every region of the C++ loop — DC symbol, AC record, a complete
// burst, the end of a block — is entered by SOME lane of a wave in nearly every iteration, so a wave executes all of them every
// time: 78 vector and 35 scalar instructions per iteration as the compiler lays them out (copies of loop-carried values around
// every region, 64-bit address arithmetic for every store, separate shift / mask / or where one instruction does).  Here an
// iteration is 58 vector instructions: the next block's tables come from a 16-byte block table read unconditionally beside the
// look-up (fill_block_table), records leave through an SGPR base + 32-bit offset, the regions are entered by writing EXEC, and the
// rare cases (a code longer than HJD_LUT_BITS, a block entry or a record that does not fit, the image's last block) sit behind
// the loop.  v64-v96 are its temporaries.  Same semantics as write_core<false, ...>, statement by statement; `mi` as SyncChain::mi.
#define KW_WIN "v64"
#define KW_A "v65"
#define KW_E "v66"
#define KW_NW "v67"
#define KW_MT "v[68:71]"
#define KW_MT0 "v68"
#define KW_MT1 "v69"
#define KW_MT2 "v70"
#define KW_R03 "v[72:75]"
#define KW_R47 "v[76:79]"
#define KW_R01 "v[72:73]"
#define KW_R23 "v[74:75]"
#define KW_R45 "v[76:77]"
#define KW_R67 "v[78:79]"
#define KW_R0 "v72"
#define KW_R1 "v73"
#define KW_LEN "v80"
#define KW_S "v81"
#define KW_ADV "v82"
#define KW_TOT "v83"
#define KW_T "v84"
#define KW_OFF "v85"
#define KW_VAL "v86"
#define KW_K "v87"
#define KW_ZW "v88"
#define KW_SLOT "v89"
#define KW_W "v90"
#define KW_RA "v91"
#define KW_TI "v92"
#define KW_REL "v93"
#define KW_EV "v94"
#define KW_EA "v95"
#define KW_GO "v96"
template <uint32_t WGSZ, uint32_t ENTCAP>
__device__ __forceinline__ void write_window_plain(uint32_t mt4_addr, uint32_t sec_addr, uint32_t slot_addr, const uint32_t *__restrict__ rowp,
                                                   uint32_t stride, uint32_t row_bit0, uint32_t &p, uint32_t &z, uint32_t &m, uint32_t end,
                                                   uint32_t &nblk, uint32_t &nrec, WriteCtx &W, uint32_t &flags)
{
    static_assert(HJD_REC_RUN == 8 && (WGSZ == 64 || WGSZ == 256), "the burst below: eight records, ring columns WGSZ words apart");
    static_assert(HJD_FLAG_BAD_CODE == 1u && HJD_FLAG_OVERRUN == 2u, "flag literals below");
    // bit reader (ReaderLds::init): hi = the word under the cursor, lo = the next one, sh = unconsumed bits of hi
    const uint32_t rel0 = p - row_bit0, k0 = rel0 >> 5, o0 = rel0 & 31u;
    const uint32_t *sp = rowp + k0 * stride;
    uint32_t hi = sp[0], lo = o0 ? sp[stride] : hi, sh = (32u - o0) & 31u;
    uint32_t nxa = lds_addr(sp + (o0 ? 2u * stride : stride));
    const uint32_t st4 = stride * 4u;
    const uint4 cur = lds_load<uint4>(mt4_addr + m * 16u);
    uint32_t acb = cur.y, tb = z ? cur.y : cur.x, mi = cur.z;
    // block and record numbers run relative to the workgroup's first inside the loop: the entry staging is indexed by them as they are
    const uint32_t b0 = W.b0, r0 = W.r0;
    uint32_t pend = end, zhi = 0u, bfirst = W.bfirst == 0xFFFFFFFFu ? 0xFFFFFFFFu : W.bfirst - r0, bdc = W.bdc, tag = W.tag;
    nblk -= b0;
    nrec -= r0; // (r0 is a multiple of HJD_REC_RUN: the workgroups' totals are sums of rounded runs)
    const uint32_t ringa = lds_addr(W.ring), enta = lds_addr(W.ent), secm = sec_addr - 2u * HJD_SEC_SIZE;
    const uint32_t reccap = W.rec_cap > r0 ? W.rec_cap - r0 : 0u, nblocks = W.nblocks - b0, b0s = b0 << 22;
    uint32_t *const recp = W.rec;
    uint2 *const blkp = W.blk;
    unsigned long long act, isdc, bm, rm, g1, g2, sv0;
    constexpr int U = WGSZ / 64; // ring columns in units of 64 words (ds_read2st64_b32)
    asm volatile(
        "s_mov_b64 %[sv0], exec\n\t"
        "s_waitcnt lgkmcnt(0)\n"
        "1:\n\t"                                                  // ---- look-up (EXEC = the lanes with bits left)
        "v_alignbit_b32 " KW_WIN ", %[hi], %[lo], %[sh]\n\t"
        "v_bfe_u32 " KW_A ", " KW_WIN ", %[lb], %[lw]\n\t"
        "v_lshl_add_u32 " KW_A ", " KW_A ", 1, %[tb]\n\t"
        "ds_read_u16 " KW_E ", " KW_A "\n\t"
        "ds_read_b32 " KW_NW ", %[nxa]\n\t"                     // next stream word
        "ds_read_b128 " KW_MT ", %[mi]\n\t"                     // next block's tables
        "v_cmp_eq_u32_e64 %[isdc], 0, %[z]\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        "v_and_b32_e32 " KW_TOT ", 31, " KW_E "\n\t"
        "v_cmp_eq_u32_e32 vcc, 0, " KW_TOT "\n\t"
        "s_and_b64 %[g1], vcc, exec\n\t"
        "s_cbranch_scc1 4f\n"                                     // a code longer than HJD_LUT_BITS
        "5:\n\t"
        // e = code length + value bits | value bits << 5 | advance << 9; value bits + DetermineSign (loadjpg.cpp:396-409)
        "v_bfe_u32 " KW_S ", " KW_E ", 5, 4\n\t"
        "v_lshrrev_b32_e32 " KW_ADV ", 9, " KW_E "\n\t"
        "v_sub_u32_e32 " KW_OFF ", 32, " KW_TOT "\n\t"
        "v_bfe_u32 " KW_VAL ", " KW_WIN ", " KW_OFF ", " KW_S "\n\t"
        "v_bfe_i32 " KW_T ", " KW_WIN ", " KW_OFF ", " KW_S "\n\t" // the same bits, sign-extended: negative = leading value bit set
        "v_lshl_add_u32 " KW_K ", -1, " KW_S ", 1\n\t"          // 1 - 2^s
        "v_cmp_gt_i32_e32 vcc, 0, " KW_T "\n\t"                 // leading value bit set: the value as it is
        "v_cndmask_b32_e64 " KW_K ", " KW_K ", 0, vcc\n\t"
        "v_add_u32_e32 " KW_VAL ", " KW_VAL ", " KW_K "\n\t"
        "v_add3_u32 " KW_ZW ", %[z], " KW_ADV ", -1\n\t"
        "v_add_u32_e32 %[z], %[z], " KW_ADV "\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, " KW_S "\n\t"
        "v_cndmask_b32_e64 %[bfirst], %[bfirst], %[nrec], %[isdc]\n\t"  // DC symbol (loadjpg.cpp:562-667): the block's entry begins here
        "v_cndmask_b32_e64 %[bdc], %[bdc], " KW_VAL ", %[isdc]\n\t"
        "v_cndmask_b32_e64 %[tb], %[tb], %[acb], %[isdc]\n\t"
        "v_cmp_lt_u32_e64 %[bm], 63, %[z]\n\t"                    // EOB or last coefficient
        "s_andn2_b64 %[rm], vcc, %[isdc]\n\t"                     // AC run/size with a value (loadjpg.cpp:776-808): a record
        "s_mov_b64 %[act], exec\n\t"
        "s_mov_b64 exec, %[rm]\n\t"                               // ---- record
        "v_and_or_b32 " KW_A ", " KW_ZW ", 63, %[slotb]\n\t"
        "ds_read_u8 " KW_SLOT ", " KW_A "\n\t"
        "v_or_b32_e32 %[zhi], %[zhi], " KW_ZW "\n\t"            // zw > 63 (bit 6): flagged after the loop
        "v_and_b32_e32 " KW_RA ", 7, %[nrec]\n\t"
        "v_cmp_eq_u32_e32 vcc, 7, " KW_RA "\n\t"
        "v_lshl_add_u32 " KW_RA ", " KW_RA ", %[lgcol], %[ringa]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_lshl_or_b32 " KW_W ", " KW_SLOT ", 16, %[tag]\n\t"
        "v_bfi_b32 " KW_W ", %[m16], " KW_VAL ", " KW_W "\n\t"
        "ds_write_b32 " KW_RA ", " KW_W "\n\t"
        "s_and_b64 exec, exec, vcc\n\t"                           // lanes whose burst is complete: whole, aligned sectors
        "s_cbranch_execz 7f\n\t"
        "v_cmp_gt_u32_e32 vcc, %[reccap], %[nrec]\n\t"
        "s_andn2_b64 %[g1], exec, vcc\n\t"
        "s_cbranch_scc1 8f\n"                                     // a burst past the image's record space
        "6:\n\t"
        "ds_read2st64_b32 " KW_R01 ", %[ringa] offset0:%[c0] offset1:%[c1]\n\t"
        "ds_read2st64_b32 " KW_R23 ", %[ringa] offset0:%[c2] offset1:%[c3]\n\t"
        "ds_read2st64_b32 " KW_R45 ", %[ringa] offset0:%[c4] offset1:%[c5]\n\t"
        "ds_read2st64_b32 " KW_R67 ", %[ringa] offset0:%[c6] offset1:%[c7]\n\t"
        "v_add_lshl_u32 " KW_GO ", %[nrec], %[r0], 2\n\t"
        "s_waitcnt lgkmcnt(2)\n\t"
        "global_store_dwordx4 " KW_GO ", " KW_R03 ", %[recp] offset:-28\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "global_store_dwordx4 " KW_GO ", " KW_R47 ", %[recp] offset:-12\n"
        "7:\n\t"
        "s_mov_b64 exec, %[rm]\n\t"
        "v_add_u32_e32 %[nrec], 1, %[nrec]\n\t"
        "s_mov_b64 exec, %[bm]\n\t"                               // ---- end of block: next block (DecodeMCU order, loadjpg.cpp:945-997)
        "v_cmp_ne_u32_e32 vcc, -1, %[bfirst]\n\t"                 // (all ones: the block began in an earlier span, no entry from here)
        "v_cmp_gt_u32_e64 %[g1], %[entcap], %[nblk]\n\t"        // (block and record numbers are relative to the workgroup's first here)
        "v_cmp_gt_u32_e64 %[g2], %[m16], %[bfirst]\n\t"
        "s_and_b64 %[g1], %[g1], %[g2]\n\t"
        "s_andn2_b64 %[g2], vcc, %[g1]\n\t"
        "s_cbranch_scc1 9f\n"                                     // more blocks or records than the staging holds: straight out
        "10:\n\t"
        "v_lshl_or_b32 " KW_EV ", %[bdc], 16, %[bfirst]\n\t"
        "v_lshl_add_u32 " KW_EA ", %[nblk], 2, %[enta]\n\t"
        "s_and_b64 exec, vcc, %[g1]\n\t"
        "ds_write_b32 " KW_EA ", " KW_EV "\n\t"
        "s_mov_b64 exec, %[bm]\n\t"
        "v_mov_b32_e32 %[bfirst], -1\n\t"
        "v_add_u32_e32 %[nblk], 1, %[nblk]\n\t"
        "v_lshl_add_u32 %[tag], %[nblk], 22, %[b0s]\n\t"
        "v_and_b32_e32 %[tag], %[tagm], %[tag]\n\t"
        "v_cmp_eq_u32_e32 vcc, %[nblocks], %[nblk]\n\t"
        "s_and_b64 %[g1], vcc, exec\n\t"
        "s_cbranch_scc1 11f\n"                                    // the image is complete
        "12:\n\t"
        "v_mov_b32_e32 %[z], 0\n\t"
        "v_mov_b32_e32 %[tb], " KW_MT0 "\n\t"
        "v_mov_b32_e32 %[acb], " KW_MT1 "\n\t"
        "v_mov_b32_e32 %[mi], " KW_MT2 "\n\t"
        "s_mov_b64 exec, %[act]\n\t"                              // ---- the bit reader moves on
        "v_sub_u32_e32 %[sh], %[sh], " KW_TOT "\n\t"
        "v_add_u32_e32 %[p], %[p], " KW_TOT "\n\t"
        "v_cmp_gt_i32_e32 vcc, 0, %[sh]\n\t"
        "v_cndmask_b32_e32 %[hi], %[hi], %[lo], vcc\n\t"
        "v_cndmask_b32_e32 %[lo], %[lo], " KW_NW ", vcc\n\t"
        "v_cndmask_b32_e32 " KW_T ", 0, %[st4], vcc\n\t"
        "v_add_u32_e32 %[nxa], %[nxa], " KW_T "\n\t"
        "v_and_b32_e32 %[sh], 31, %[sh]\n\t"
        "v_cmp_lt_u32_e32 vcc, %[p], %[pend]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_branch 3f\n"
        "4:\n\t"                                                  // ---- second-level table on the next HJD_SEC_BITS bits (IsInHuffmanCodes,
        "s_mov_b64 %[act], exec\n\t"                              //      loadjpg.cpp:335-392, for the lengths the primary table cannot hold)
        "s_mov_b64 exec, %[g1]\n\t"
        "v_lshrrev_b32_e32 " KW_T ", 5, " KW_E "\n\t"           // 1 + table index, 0 = no code has this prefix
        "v_bfe_u32 " KW_A ", " KW_WIN ", 16, %[sw]\n\t"
        "v_lshl_add_u32 " KW_A ", " KW_T ", %[sw], " KW_A "\n\t"
        "v_lshl_add_u32 " KW_A ", " KW_A ", 1, %[secm]\n\t"
        "ds_read_u16 " KW_A ", " KW_A "\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, " KW_T "\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cndmask_b32_e32 " KW_E ", 0, " KW_A ", vcc\n\t"
        "v_and_b32_e32 " KW_TOT ", 31, " KW_E "\n\t"
        "v_cmp_eq_u32_e32 vcc, 0, " KW_TOT "\n\t"
        "v_cndmask_b32_e64 " KW_T ", 0, 1, vcc\n\t"
        "v_or_b32_e32 %[flags], %[flags], " KW_T "\n\t"         // HJD_FLAG_BAD_CODE
        "v_cndmask_b32_e64 " KW_E ", " KW_E ", 16, vcc\n\t"     // skip 16 bits: any progress will do, the result is flagged
        "v_and_b32_e32 " KW_TOT ", 31, " KW_E "\n\t"
        "s_mov_b64 exec, %[act]\n\t"
        "s_branch 5b\n"
        "8:\n\t"                                                  // ---- bursts past the record space are not stored (g1: those lanes)
        "s_mov_b64 %[g2], exec\n\t"
        "s_mov_b64 exec, %[g1]\n\t"
        "v_or_b32_e32 %[flags], 2, %[flags]\n\t"                  // HJD_FLAG_OVERRUN
        "s_andn2_b64 exec, %[g2], %[g1]\n\t"
        "s_branch 6b\n"
        "9:\n\t"                                                  // ---- block entries that do not fit the staging (g2: those lanes)
        "s_mov_b64 exec, %[g2]\n\t"
        "v_add_u32_e32 " KW_R0 ", %[bfirst], %[r0]\n\t"
        "v_and_b32_e32 " KW_R1 ", %[m16], %[bdc]\n\t"
        "v_add_lshl_u32 " KW_GO ", %[nblk], %[b0], 3\n\t"
        "s_nop 0\n\t"
        "global_store_dwordx2 " KW_GO ", " KW_R01 ", %[blkp]\n\t"
        "s_mov_b64 exec, %[bm]\n\t"
        "s_branch 10b\n"
        "11:\n\t"                                                 // ---- the last block (g1: those lanes): end of its records; what follows is not looked at
        "s_mov_b64 exec, %[g1]\n\t"
        "v_add_u32_e32 " KW_R0 ", %[nrec], %[r0]\n\t"
        "v_mov_b32_e32 " KW_R1 ", 0\n\t"
        "v_add_lshl_u32 " KW_GO ", %[nblk], %[b0], 3\n\t"
        "v_add_u32_e32 %[pend], %[p], " KW_TOT "\n\t"
        "global_store_dwordx2 " KW_GO ", " KW_R01 ", %[blkp]\n\t"
        "s_mov_b64 exec, %[bm]\n\t"
        "s_branch 12b\n"
        "3:\n\t"
        "s_mov_b64 exec, %[sv0]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [p] "+v"(p), [z] "+v"(z), [sh] "+v"(sh), [hi] "+v"(hi), [lo] "+v"(lo), [nxa] "+v"(nxa), [tb] "+v"(tb), [acb] "+v"(acb), [mi] "+v"(mi),
          [nblk] "+v"(nblk), [nrec] "+v"(nrec), [bfirst] "+v"(bfirst), [bdc] "+v"(bdc), [tag] "+v"(tag), [flags] "+v"(flags), [zhi] "+v"(zhi),
          [pend] "+v"(pend), [act] "=&s"(act), [isdc] "=&s"(isdc), [bm] "=&s"(bm), [rm] "=&s"(rm), [g1] "=&s"(g1), [g2] "=&s"(g2), [sv0] "=&s"(sv0)
        : [ringa] "v"(ringa), [enta] "v"(enta), [secm] "v"(secm), [slotb] "v"(slot_addr), [st4] "v"(st4), [b0] "v"(b0), [r0] "v"(r0), [b0s] "v"(b0s),
          [reccap] "s"(reccap), [nblocks] "s"(nblocks), [entcap] "s"(ENTCAP), [m16] "s"(0xFFFFu), [tagm] "s"(0x1FC00000u), [recp] "s"(recp),
          [blkp] "s"(blkp), [lb] "n"(32 - HJD_LUT_BITS), [lw] "n"(HJD_LUT_BITS), [sw] "n"(HJD_SEC_BITS), [lgcol] "n"(WGSZ == 256 ? 10 : 8),
          [c0] "n"(0 * U), [c1] "n"(1 * U), [c2] "n"(2 * U), [c3] "n"(3 * U), [c4] "n"(4 * U), [c5] "n"(5 * U), [c6] "n"(6 * U), [c7] "n"(7 * U)
        : "vcc", "scc", "memory", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
          "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96");
    flags |= (zhi >> 5) & HJD_FLAG_OVERRUN; // zw <= 63 + 15, so bit 6 says it
    m = lds_load<uint32_t>(mi + 12u);       // the block in progress
    nblk += b0;
    nrec += r0;
    W.bfirst = bfirst == 0xFFFFFFFFu ? 0xFFFFFFFFu : bfirst + r0;
    W.bdc = bdc;
    W.tag = tag;
}

// All chains of a lane through one staging window: until every chain's bit position has reached pend[c] (relative to the
// window's first bit; 0 = the chain takes no part).  sec_addr: LDS byte address of the second-level tables.
template <int NC>
__device__ __forceinline__ void sync_window(SyncChain (&ch)[NC], const uint32_t (&pend)[NC], uint32_t sec_addr)
{
    uint32_t plim[NC]; // two-symbol steps stop 32 bits short of the window's end (see decode_windows)
#pragma unroll
    for (int c = 0; c < NC; c++) plim[c] = pend[c] > 32u ? pend[c] - 32u : 0u;
    for (;;) {
        unsigned long long actm[NC], anym = 0ull;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            actm[c] = __builtin_amdgcn_ballot_w64(ch[c].p < pend[c]);
            anym |= actm[c];
        }
        if (anym == 0ull) break; // wave-uniform: EXEC stays full inside the loop
        uint32_t win[NC], nw[NC], e[NC], w2[NC];
        uint4 mt[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            win[c] = __builtin_amdgcn_alignbit(ch[c].hi, ch[c].lo, ch[c].sh);
            const uint32_t tb = ch[c].z == 0u ? ch[c].dcb : ch[c].acb;
            w2[c] = lds_load<uint32_t>(tb + ((win[c] >> (32 - HJD_LUT_BITS_S)) << 2)); // [0] | [1] << 16
            nw[c] = lds_load<uint32_t>(ch[c].nxa);
            mt[c] = lds_load<uint4>(ch[c].mi);
        }
        unsigned long long lngm = 0ull;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            // [0] this symbol, [1] this symbol and the next as one step.  The pair is taken while the first symbol cannot end the
            // block (an AC advance is at most 16) and the second still starts inside the window (a step is at most 31 bits):
            // the states passed on are then exactly those of the symbol-by-symbol walk of the final pass.
            const bool pair = ch[c].z < 48u && ch[c].p < plim[c];
            e[c] = pair ? w2[c] >> 16 : w2[c] & 0xFFFFu;
            lngm |= actm[c] & __builtin_amdgcn_ballot_w64((e[c] & 31u) == 0u);
        }
        if (lngm != 0ull) {
            // code longer than HJD_LUT_BITS: second-level table on the next HJD_SEC_BITS bits (IsInHuffmanCodes,
            // loadjpg.cpp:335-392, for the lengths the primary table cannot hold); 0 = no such code
#pragma unroll
            for (int c = 0; c < NC; c++)
                if ((e[c] & 31u) == 0u) {
                    const uint32_t sub = e[c] >> 5;
                    e[c] = sub ? lds_load<uint16_t>(sec_addr + (((sub - 1u) * HJD_SEC_SIZE_S + ((win[c] >> 16) & (HJD_SEC_SIZE_S - 1u))) << 1)) : 0u;
                    if ((e[c] & 31u) == 0u) e[c] = 16u; // skip 16 bits: any progress will do, the final pass flags it
                }
        }
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const uint32_t ev = ch[c].p < pend[c] ? e[c] : 0u; // a chain at the end of its window stands still
            const uint32_t tot = ev & 31u;
            const uint32_t z2 = ch[c].z + __builtin_amdgcn_ubfe(ev, 5, 7);
            const bool blk_end = z2 > 63u; // EOB or the last coefficient: next block (DecodeMCU order, loadjpg.cpp:945-997)
            ch[c].nrec += ev >> 14;
            ch[c].nblk += blk_end ? 1u : 0u;
            ch[c].z = blk_end ? 0u : z2;
            ch[c].dcb = blk_end ? mt[c].x : ch[c].dcb;
            ch[c].acb = blk_end ? mt[c].y : ch[c].acb;
            ch[c].mi = blk_end ? mt[c].z : ch[c].mi;
            ch[c].p += tot;
            const int sh2 = (int)ch[c].sh - (int)tot;
            const bool roll = sh2 < 0;
            ch[c].hi = roll ? ch[c].lo : ch[c].hi;
            ch[c].lo = roll ? nw[c] : ch[c].lo;
            ch[c].nxa += (uint32_t)sh2 & 0x100u; // + 256 bytes (the next word of the column) when the reader rolls: sh2 in -31..-1
            ch[c].sh = (uint32_t)sh2 & 31u;
        }
    }
}

// One chain per lane (HJD_SYNC_C = 1, and every sparse refinement round): the same loop, instruction by instruction.  A wave issues
// one instruction of any kind every 4-5 cycles whatever the other waves do, so the time of a walk is its instruction count plus
// the LDS round trip of the table look-up that every step waits for: the compiler's version of the loop above is 58
// instructions per step (copies for its loop-carried values, nine for the wave-uniform test of a 0.2 % case, separate shift /
// mask / add for what v_bfe_u32 and v_lshl_add_u32 do in one) and waits with nothing else to do.  Here a step is 40
// instructions, and the ones that do not feed the next look-up (record and block counts, the bit position and the test for the
// window's end, the conditions for a two-symbol step) are the PREVIOUS step's, issued between the look-up and the wait for it.
// v120-v123 receive the block table entry (ds_read_b128 needs four consecutive registers; inline assembly cannot name the
// parts of a 128-bit operand).  All waits for the LDS reads issued here are inside: the compiler does not count them.
template <>
__device__ __forceinline__ void sync_window<1>(SyncChain (&ch)[1], const uint32_t (&pend)[1], uint32_t sec_addr)
{
    SyncChain &c = ch[0];
    const uint32_t plim = pend[0] > 32u ? pend[0] - 32u : 0u;
    const uint32_t secm = sec_addr - 2u * HJD_SEC_SIZE_S; // second-level table `sub` (1-based) starts at sec_addr + (sub - 1) * 2 * HJD_SEC_SIZE_S
    uint32_t tb = c.z == 0u ? c.dcb : c.acb;            // table of the next symbol: carried from step to step
    uint32_t win, a, w, nw, t;
    uint32_t e = 0u, tot = 0u;                          // the previous step's entry and length: its counts are added a step late
    unsigned long long act, pm, sv, bm = 0ull;
    static_assert(HJD_LUT_BITS_S >= 8 && HJD_LUT_BITS_S <= 12 && HJD_SEC_BITS_S == 16 - HJD_LUT_BITS_S, "field positions below: the %[lb] / %[lw] / %[sw] immediates");
    asm volatile(
        // One taken branch costs a wave as much as four instructions (tools/lat_probe.hip: 30 cycles against 8), so the loop is laid
        // out with a single one per step: its back edge is the test for the window's end, and the second-level look-up (0.2 % of
        // the symbols) sits behind the loop, reached by a branch that is not taken otherwise.
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_branch 1f\n"
        "2:\n\t"                                             // ---- the look-up has been issued: is the step taken at all?
        "s_and_b64 vcc, vcc, %[pm]\n\t"                    // two symbols as one step
        "s_waitcnt lgkmcnt(2)\n\t"
        "v_lshrrev_b32_e32 %[e], 16, %[w]\n\t"
        "v_cndmask_b32_e32 %[e], %[w], %[e], vcc\n\t"      // bits 16.. of e are not looked at
        "v_and_b32_e32 %[tot], 31, %[e]\n\t"
        "v_cmp_eq_u32_e32 vcc, 0, %[tot]\n\t"
        "s_and_b64 %[pm], vcc, %[act]\n\t"                 // a live lane with a code longer than HJD_LUT_BITS?
        "s_cbranch_scc1 4f\n"
        "5:\n\t"
        "v_cndmask_b32_e64 %[e], 0, %[e], %[act]\n\t"      // a lane at the end of its window stands still
        "v_and_b32_e32 %[tot], 31, %[e]\n\t"
        "v_bfe_u32 %[t], %[e], 5, 7\n\t"
        "v_sub_u32_e32 %[sh], %[sh], %[tot]\n\t"
        "v_add_u32_e32 %[z], %[z], %[t]\n\t"
        "v_cmp_gt_i32_e32 vcc, 0, %[sh]\n\t"               // the reader rolls on to the next word
        "v_cmp_lt_u32_e64 %[bm], 63, %[z]\n\t"             // end of block
        "v_and_b32_e32 %[t], 0x100, %[sh]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cndmask_b32_e32 %[hi], %[hi], %[lo], vcc\n\t"
        "v_add_u32_e32 %[nxa], %[nxa], %[t]\n\t"
        "v_cndmask_b32_e64 %[tb], %[acb], v120, %[bm]\n\t" // the next symbol: this block's AC table, or the next block's DC table
        "v_cndmask_b32_e32 %[lo], %[lo], %[nw], vcc\n\t"
        "v_cndmask_b32_e64 %[z], %[z], 0, %[bm]\n\t"
        "v_and_b32_e32 %[sh], 31, %[sh]\n\t"
        "v_cndmask_b32_e64 %[mi], %[mi], v122, %[bm]\n\t"
        "v_cndmask_b32_e64 %[acb], %[acb], v121, %[bm]\n"
        "1:\n\t"                                             // ---- next look-up
        "v_alignbit_b32 %[win], %[hi], %[lo], %[sh]\n\t"   // the next 32 bits
        "v_bfe_u32 %[a], %[win], %[lb], %[lw]\n\t"       // the next HJD_LUT_BITS_S bits
        "v_lshl_add_u32 %[a], %[a], 2, %[tb]\n\t"
        "ds_read_b32 %[w], %[a]\n\t"                       // [0] | [1] << 16
        "ds_read_b32 %[nw], %[nxa]\n\t"                    // next stream word
        "ds_read_b128 v[120:123], %[mi]\n\t"               // next block's tables
        // while the look-up is under way: what the previous step left to do
        "v_add_u32_e32 %[p], %[p], %[tot]\n\t"
        "v_bfe_u32 %[t], %[e], 14, 2\n\t"
        "v_addc_co_u32_e64 %[nblk], %[sv], 0, %[nblk], %[bm]\n\t"
        "v_cmp_lt_u32_e64 %[act], %[p], %[pend]\n\t"       // lanes with bits left in the window
        "v_add_u32_e32 %[nrec], %[nrec], %[t]\n\t"
        "v_cmp_gt_u32_e64 %[pm], 48, %[z]\n\t"
        "v_cmp_lt_u32_e32 vcc, %[p], %[plim]\n\t"
        "s_cmp_lg_u64 %[act], 0\n\t"
        "s_cbranch_scc1 2b\n\t"                            // the one taken branch of a step
        "s_branch 3f\n"
        "4:\n\t"                                             // ---- second-level table on the next HJD_SEC_BITS bits
        "s_and_saveexec_b64 %[sv], %[pm]\n\t"
        "v_bfe_u32 %[t], %[e], 5, 11\n\t"                  // 1 + table index, 0 = no code has this prefix
        "v_bfe_u32 %[a], %[win], 16, %[sw]\n\t"
        "v_lshl_add_u32 %[a], %[t], %[sw], %[a]\n\t"
        "v_lshl_add_u32 %[a], %[a], 1, %[secm]\n\t"
        "ds_read_u16 %[a], %[a]\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %[t]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cndmask_b32_e32 %[e], 0, %[a], vcc\n\t"
        "v_and_b32_e32 %[tot], 31, %[e]\n\t"
        "v_cmp_eq_u32_e32 vcc, 0, %[tot]\n\t"
        "v_cndmask_b32_e64 %[e], %[e], 16, vcc\n\t"        // no such code: skip 16 bits, the final pass flags it
        "s_mov_b64 exec, %[sv]\n\t"
        "s_branch 5b\n"
        "3:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [p] "+v"(c.p), [z] "+v"(c.z), [sh] "+v"(c.sh), [hi] "+v"(c.hi), [lo] "+v"(c.lo), [nxa] "+v"(c.nxa), [tb] "+v"(tb),
          [acb] "+v"(c.acb), [mi] "+v"(c.mi), [nblk] "+v"(c.nblk), [nrec] "+v"(c.nrec), [e] "+v"(e), [tot] "+v"(tot), [bm] "+s"(bm),
          [win] "=&v"(win), [a] "=&v"(a), [w] "=&v"(w), [t] "=&v"(t), [nw] "=&v"(nw), [act] "=&s"(act), [pm] "=&s"(pm), [sv] "=&s"(sv)
        : [pend] "v"(pend[0]), [plim] "v"(plim), [secm] "v"(secm), [lb] "n"(32 - HJD_LUT_BITS_S), [lw] "n"(HJD_LUT_BITS_S), [sw] "n"(HJD_SEC_BITS_S)
        : "vcc", "scc", "memory", "v120", "v121", "v122", "v123");
}

// The same loop for restart streams (images with DRI; round 5).  Rounds 3-4 ran the C++ loops on them; the first form of this round
// ran the plain hand-written loop up to 39 bits before every marker and the C++ rule loop around it, in alternating phases — but a
// wave then walks a window up to twice (the lanes that met a marker finish their window while the others wait): 64 x 1080p with one
// marker per MCU row 152 Gpixel/s against 182 without DRI.  Here the speculative form of the reference's marker rule
// (loadjpg.cpp:535-550: its check fires at a block start when the byte cursor sits on the marker, i.e. when a block has ended inside
// the last byte before it — or the decode has run over the marker altogether) is PART of the loop: three instructions in the shadow
// of the look-up compare the position after the previous step (+ 8 when it ended a block) with the next marker's, and a lane that
// trips is pulled onto the interval's start out of line (block 6: position, block table entry 0, bit reader, next marker), after
// which every lane repeats its look-up.  Two-symbol steps stop 40 bits before the marker, as in sync_core<true>.
// mrel / mrel2: positions of the next two markers relative to the window's first bit (0x7FFFFFFF: none); a lane that needs a third
// one inside the window stops (its pend becomes its position) and the caller hands it the next two.  col: LDS byte address of the
// lane's staging column; mt0: of block table entry 0.
__device__ __forceinline__ void sync_window_rst(SyncChain &c, uint32_t &pend, uint32_t &mrel, uint32_t &mrel2, uint32_t col, uint32_t mt0,
                                                uint32_t sec_addr)
{
    const uint32_t plimw = pend > 32u ? pend - 32u : 0u;
    uint32_t plim = min(plimw, max(mrel, 40u) - 40u);
    const uint32_t secm = sec_addr - 2u * HJD_SEC_SIZE_S;
    uint32_t tb = c.z == 0u ? c.dcb : c.acb;
    uint32_t win, a, w, nw, t, t2;
    uint32_t e = 0u, tot = 0u;
    unsigned long long act, pm, sv, jm, bm = 0ull;
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_branch 1f\n"
        "2:\n\t"                                             // ---- the look-up has been issued: is the step taken at all?
        "s_and_b64 vcc, vcc, %[pm]\n\t"                    // two symbols as one step
        "s_waitcnt lgkmcnt(2)\n\t"
        "v_lshrrev_b32_e32 %[e], 16, %[w]\n\t"
        "v_cndmask_b32_e32 %[e], %[w], %[e], vcc\n\t"
        "v_and_b32_e32 %[tot], 31, %[e]\n\t"
        "v_cmp_eq_u32_e32 vcc, 0, %[tot]\n\t"
        "s_and_b64 %[pm], vcc, %[act]\n\t"                 // a live lane with a code longer than HJD_LUT_BITS_S?
        "s_cbranch_scc1 4f\n"
        "5:\n\t"
        "v_cndmask_b32_e64 %[e], 0, %[e], %[act]\n\t"      // a lane at the end of its window stands still
        "v_and_b32_e32 %[tot], 31, %[e]\n\t"
        "v_bfe_u32 %[t], %[e], 5, 7\n\t"
        "v_sub_u32_e32 %[sh], %[sh], %[tot]\n\t"
        "v_add_u32_e32 %[z], %[z], %[t]\n\t"
        "v_cmp_gt_i32_e32 vcc, 0, %[sh]\n\t"
        "v_cmp_lt_u32_e64 %[bm], 63, %[z]\n\t"             // end of block
        "v_and_b32_e32 %[t], 0x100, %[sh]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cndmask_b32_e32 %[hi], %[hi], %[lo], vcc\n\t"
        "v_add_u32_e32 %[nxa], %[nxa], %[t]\n\t"
        "v_cndmask_b32_e64 %[tb], %[acb], v120, %[bm]\n\t"
        "v_cndmask_b32_e32 %[lo], %[lo], %[nw], vcc\n\t"
        "v_cndmask_b32_e64 %[z], %[z], 0, %[bm]\n\t"
        "v_and_b32_e32 %[sh], 31, %[sh]\n\t"
        "v_cndmask_b32_e64 %[mi], %[mi], v122, %[bm]\n\t"
        "v_cndmask_b32_e64 %[acb], %[acb], v121, %[bm]\n"
        "1:\n\t"                                             // ---- next look-up
        "v_alignbit_b32 %[win], %[hi], %[lo], %[sh]\n\t"
        "v_bfe_u32 %[a], %[win], %[lb], %[lw]\n\t"
        "v_lshl_add_u32 %[a], %[a], 2, %[tb]\n\t"
        "ds_read_b32 %[w], %[a]\n\t"
        "ds_read_b32 %[nw], %[nxa]\n\t"
        "ds_read_b128 v[120:123], %[mi]\n\t"
        // while the look-up is under way: what the previous step left to do, and the marker rule on the position it reached
        "v_add_u32_e32 %[p], %[p], %[tot]\n\t"
        "v_cndmask_b32_e64 %[t2], 0, 8, %[bm]\n\t"
        "v_bfe_u32 %[t], %[e], 14, 2\n\t"
        "v_add_u32_e32 %[t2], %[t2], %[p]\n\t"
        "v_addc_co_u32_e64 %[nblk], %[sv], 0, %[nblk], %[bm]\n\t"
        "v_cmp_gt_u32_e64 %[jm], %[t2], %[mrel]\n\t"       // block ended within the marker's last byte, or the decode ran over the marker
        "v_cmp_lt_u32_e64 %[act], %[p], %[pend]\n\t"
        "v_add_u32_e32 %[nrec], %[nrec], %[t]\n\t"
        "s_cmp_lg_u64 %[jm], 0\n\t"
        "s_cbranch_scc1 6f\n\t"
        "v_cmp_gt_u32_e64 %[pm], 48, %[z]\n\t"
        "v_cmp_lt_u32_e32 vcc, %[p], %[plim]\n\t"
        "s_cmp_lg_u64 %[act], 0\n\t"
        "s_cbranch_scc1 2b\n\t"
        "s_branch 3f\n"
        "4:\n\t"                                             // ---- second-level table on the next HJD_SEC_BITS_S bits
        "s_and_saveexec_b64 %[sv], %[pm]\n\t"
        "v_bfe_u32 %[t], %[e], 5, 11\n\t"
        "v_bfe_u32 %[a], %[win], 16, %[sw]\n\t"
        "v_lshl_add_u32 %[a], %[t], %[sw], %[a]\n\t"
        "v_lshl_add_u32 %[a], %[a], 1, %[secm]\n\t"
        "ds_read_u16 %[a], %[a]\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %[t]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cndmask_b32_e32 %[e], 0, %[a], vcc\n\t"
        "v_and_b32_e32 %[tot], 31, %[e]\n\t"
        "v_cmp_eq_u32_e32 vcc, 0, %[tot]\n\t"
        "v_cndmask_b32_e64 %[e], %[e], 16, vcc\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_branch 5b\n"
        "6:\n\t"                                             // ---- lanes %[jm] are pulled onto the start of the next interval
        "s_waitcnt lgkmcnt(0)\n\t"                         // the look-ups issued above land before their registers are used again
        "s_and_saveexec_b64 %[sv], %[jm]\n\t"
        "v_mov_b32_e32 %[p], %[mrel]\n\t"                  // position = the marker's (the interval's first bit), block start, block 0 of an MCU
        "v_mov_b32_e32 %[z], 0\n\t"
        "ds_read_b128 v[120:123], %[mt0]\n\t"              // block table entry 0: DC table, AC table, entry of the next block
        "v_min_u32_e32 %[a], %[p], %[pend]\n\t"            // (a jump past the window's end: the lane is done, its reader only has to stay inside the staged words)
        "v_lshrrev_b32_e32 %[a], 5, %[a]\n\t"
        "v_and_b32_e32 %[t2], 31, %[p]\n\t"                // bit offset inside the word
        "v_lshl_add_u32 %[a], %[a], 8, %[col]\n\t"         // the word's LDS address (256 bytes between the words of a column)
        "ds_read_b32 %[hi], %[a]\n\t"
        "ds_read_b32 %[t], %[a] offset:256\n\t"
        "v_sub_u32_e32 %[sh], 32, %[t2]\n\t"
        "v_min_u32_e32 %[t2], 1, %[t2]\n\t"                // 1 when the position is inside a word: hi, lo = this word, the next
        "v_and_b32_e32 %[sh], 31, %[sh]\n\t"
        "v_lshl_add_u32 %[t2], %[t2], 8, %[a]\n\t"
        "v_add_u32_e32 %[nxa], 0x100, %[t2]\n\t"           // the word after hi (position at a word's start: lo = hi) or after lo
        "v_mov_b32_e32 %[mrel], %[mrel2]\n\t"              // the marker after that one; the one after THAT is not known here
        "v_mov_b32_e32 %[mrel2], 0x7ffffffe\n\t"
        "v_max_u32_e32 %[t2], 40, %[mrel]\n\t"
        "v_subrev_u32_e32 %[t2], 40, %[t2]\n\t"
        "v_min_u32_e32 %[plim], %[plimw], %[t2]\n\t"       // two-symbol steps up to 40 bits before it
        "v_cmp_eq_u32_e32 vcc, 0x7ffffffe, %[mrel]\n\t"
        "v_cndmask_b32_e32 %[pend], %[pend], %[p], vcc\n\t" // a third marker inside this window: the lane stops here, the caller goes on
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %[sh]\n\t"               // (sh = 0 <=> bit offset 0: lo = hi then)
        "v_cndmask_b32_e32 %[lo], %[hi], %[t], vcc\n\t"
        "v_mov_b32_e32 %[tb], v120\n\t"
        "v_mov_b32_e32 %[acb], v121\n\t"
        "v_mov_b32_e32 %[mi], v122\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "v_mov_b32_e32 %[e], 0\n\t"                        // nothing is left over from the step before (every lane: its counts have been added)
        "v_mov_b32_e32 %[tot], 0\n\t"
        "s_mov_b64 %[bm], 0\n\t"
        "s_branch 1b\n"
        "3:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : [p] "+v"(c.p), [z] "+v"(c.z), [sh] "+v"(c.sh), [hi] "+v"(c.hi), [lo] "+v"(c.lo), [nxa] "+v"(c.nxa), [tb] "+v"(tb),
          [acb] "+v"(c.acb), [mi] "+v"(c.mi), [nblk] "+v"(c.nblk), [nrec] "+v"(c.nrec), [e] "+v"(e), [tot] "+v"(tot), [bm] "+s"(bm),
          [pend] "+v"(pend), [mrel] "+v"(mrel), [mrel2] "+v"(mrel2), [plim] "+v"(plim),
          [win] "=&v"(win), [a] "=&v"(a), [w] "=&v"(w), [t] "=&v"(t), [t2] "=&v"(t2), [nw] "=&v"(nw), [act] "=&s"(act), [pm] "=&s"(pm), [sv] "=&s"(sv),
          [jm] "=&s"(jm)
        : [plimw] "v"(plimw), [secm] "v"(secm), [col] "v"(col), [mt0] "v"(mt0), [lb] "n"(32 - HJD_LUT_BITS_S), [lw] "n"(HJD_LUT_BITS_S), [sw] "n"(HJD_SEC_BITS_S)
        : "vcc", "scc", "memory", "v120", "v121", "v122", "v123");
}

// What the decode of a window needs besides the chains' own state.
struct RowEnv {
    const HjdParams *P;
    const HjdImage *im;
    EntropyLds L;           // restart streams: the one-chain loop with the marker rules
    uint32_t tabY, tabCb, tabCr;
    const uint32_t *wbase;  // the image's transposed stream
    uint32_t *s_stage;      // this WAVE's staging windows: chain c's column of this lane at s_stage[c * (WPP + 2) * 64 + lane + k * 64]
    uint32_t mt4_addr;      // LDS byte address of the block table (fill_block_table)
    uint32_t sec_addr;      // LDS byte address of the second-level tables
    uint32_t total_bits;
    uint32_t nmark;         // restart mode: markers found
};

// Staging window ph[c] of subsequence row[c], for every live chain c of this lane: stage WPP + 2 words of the row into the chain's
// LDS column and decode from state (p, z, m) until the window's end; nblk / nrec = blocks completed / AC records inside the
// window.  Chains with p already past the window's end (a marker jump) pass through unchanged.
// Two-symbol steps stop 32 bits short of the WINDOW's end: the state a decode holds at a window's end is then the first symbol
// boundary at or past it whatever pairs were taken before — the states passed on at a window's end are those of the
// symbol-by-symbol final pass, and two decodes that have fallen into step show the same state at every window's end (the
// checkpoint comparison of sync_body relies on that).
// `mk` (restart streams): in, an index at or before the first marker past p (0xFFFFFFFF: none known, search for it); out, the
// scan's position after the window — a walk carries it from window to window (a binary search per window was 13 dependent
// global loads for an image with 8000 markers).
template <int NC>
__device__ __forceinline__ void decode_windows(const RowEnv &E, const bool (&live)[NC], const uint32_t (&row)[NC], const uint32_t (&ph)[NC],
                                               uint32_t (&p)[NC], uint32_t (&z)[NC], uint32_t (&m)[NC], uint32_t (&nblk)[NC],
                                               uint32_t (&nrec)[NC], uint32_t (&mk)[NC])
{
    const HjdParams &P = *E.P;
    const HjdImage &im = *E.im;
    const uint32_t S = P.subseq_bits, lw = P.log2_words, WPP = P.phase_words;
    const uint32_t lane = threadIdx.x & 63u;
    bool go[NC];
    uint32_t bit0[NC], pend[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const uint32_t sp = row[c] * S;
        const uint32_t end = min(sp + S, E.total_bits);
        bit0[c] = sp + ph[c] * WPP * 32u;
        pend[c] = min(bit0[c] + WPP * 32u, end);
        go[c] = live[c] && p[c] < pend[c] && p[c] >= bit0[c];
        nblk[c] = 0u;
        nrec[c] = 0u;
    }
    // stage: all loads of all chains first (one global round trip), then the LDS stores
    constexpr uint32_t MAXW = 66u;
    (void)MAXW;
#pragma unroll
    for (int c = 0; c < NC; c++) {
        if (go[c]) {
            const uint32_t *src = row_base(E.wbase, row[c], lw) + (size_t)ph[c] * WPP * 64u;
            uint32_t *col = E.s_stage + (uint32_t)c * (WPP + 2u) * 64u + lane;
            for (uint32_t k = 0; k < WPP + 2u; k++) col[k * 64u] = src[k * 64u];
        }
    }
    if (im.restart && !im.conformant) {
        // Restart streams: the hand-written loop with the marker rule inside (sync_window_rst).  The lane brings the next two markers
        // past its position, relative to the window; one that meets a third marker inside the window comes back early and is sent
        // in again with the next two.
        const uint32_t *mpos = P.mpos + im.mark_off;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (go[c] && mk[c] == 0xFFFFFFFFu) {
                uint32_t lo = 0, hi = E.nmark; // first marker past the state's position
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (mpos[mid] <= p[c]) lo = mid + 1;
                    else hi = mid;
                }
                mk[c] = lo;
            }
            const uint32_t col = lds_addr(E.s_stage + (uint32_t)c * (WPP + 2u) * 64u + lane);
            bool left = go[c];
            while (__any(left)) {
                uint32_t mrel = 0x7FFFFFFFu, mrel2 = 0x7FFFFFFFu, prel = 0u;
                SyncChain ch;
                if (left) {
                    while (mk[c] < E.nmark && mpos[mk[c]] <= p[c]) mk[c]++;
                    if (mk[c] < E.nmark) mrel = min(mpos[mk[c]] - bit0[c], 0x7FFFFFF0u);
                    if (mk[c] + 1u < E.nmark) mrel2 = min(mpos[mk[c] + 1u] - bit0[c], 0x7FFFFFF0u);
                    prel = pend[c] - bit0[c];
                }
                const uint32_t rel = left ? p[c] - bit0[c] : 0u;
                const uint4 me = lds_load<uint4>(E.mt4_addr + (left ? m[c] : 0u) * 16u);
                ch.p = rel;
                ch.z = left ? z[c] : 0u;
                ch.dcb = me.x;
                ch.acb = me.y;
                ch.mi = me.z;
                ch.nblk = 0u;
                ch.nrec = 0u;
                const uint32_t k0 = rel >> 5, o0 = rel & 31u;
                const uint32_t wa = lds_load<uint32_t>(col + k0 * 256u), wb = lds_load<uint32_t>(col + k0 * 256u + 256u);
                ch.hi = wa;
                ch.lo = o0 ? wb : wa;
                ch.nxa = col + (k0 + (o0 ? 2u : 1u)) * 256u;
                ch.sh = (32u - o0) & 31u;
                uint32_t pstop = prel;
                sync_window_rst(ch, pstop, mrel, mrel2, col, E.mt4_addr, E.sec_addr);
                if (left) {
                    p[c] = bit0[c] + ch.p;
                    z[c] = ch.z;
                    m[c] = lds_load<uint4>(ch.mi).w;
                    nblk[c] += ch.nblk;
                    nrec[c] += ch.nrec;
                    left = p[c] < pend[c]; // stopped at a third marker: once more, with the markers after it
                }
            }
            if (go[c]) {
                while (mk[c] < E.nmark && mpos[mk[c]] <= p[c]) mk[c]++;
            }
        }
        return;
    }
    if (im.restart) { // conformant mode (never the default): the standard's interval rule, the C++ loops
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (!go[c]) continue;
            const uint32_t *col = E.s_stage + (uint32_t)c * (WPP + 2u) * 64u + lane;
            RestartCtx rc = {P.mpos + im.mark_off, E.nmark, 0u, im.ri_blocks, 0u, im.conformant};
            if (mk[c] == 0xFFFFFFFFu) {
                uint32_t lo = 0, hi = rc.nmark; // first marker past the state's position
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (rc.mpos[mid] <= p[c]) lo = mid + 1;
                    else hi = mid;
                }
                mk[c] = lo;
            }
            rc.mk = min(mk[c], rc.nmark); // sync_span moves it past p
            const uint32_t pair_lim = pend[c] > 32u ? pend[c] - 32u : 0u;
            sync_span<true>(E.L, col, bit0[c], E.tabY, E.tabCb, E.tabCr, im.nby, im.nb, p[c], z[c], m[c], pend[c], pair_lim, nblk[c], nrec[c], rc);
            mk[c] = rc.mk;
        }
        return;
    }
    SyncChain ch[NC];
    uint32_t prel[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const uint32_t col = lds_addr(E.s_stage + (uint32_t)c * (WPP + 2u) * 64u + lane);
        const uint32_t rel = go[c] ? p[c] - bit0[c] : 0u;
        const uint4 me = lds_load<uint4>(E.mt4_addr + (go[c] ? m[c] : 0u) * 16u);
        ch[c].p = rel;
        ch[c].z = go[c] ? z[c] : 0u;
        ch[c].dcb = me.x;
        ch[c].acb = me.y;
        ch[c].mi = me.z;
        ch[c].nblk = 0u;
        ch[c].nrec = 0u;
        // bit reader (see ReaderLds): hi, lo and sh = number of unconsumed bits of hi (0..31); window = alignbit(hi, lo, sh)
        const uint32_t k0 = rel >> 5, o0 = rel & 31u;
        const uint32_t a = lds_load<uint32_t>(col + k0 * 256u), b = lds_load<uint32_t>(col + k0 * 256u + 256u);
        ch[c].hi = a;
        ch[c].lo = o0 ? b : a;
        ch[c].nxa = col + (k0 + (o0 ? 2u : 1u)) * 256u;
        ch[c].sh = (32u - o0) & 31u;
        prel[c] = go[c] ? pend[c] - bit0[c] : 0u;
    }
    sync_window<NC>(ch, prel, E.sec_addr);
#pragma unroll
    for (int c = 0; c < NC; c++) {
        if (go[c]) {
            p[c] = bit0[c] + ch[c].p;
            z[c] = ch[c].z;
            m[c] = lds_load<uint4>(ch[c].mi).w;
            nblk[c] = ch[c].nblk;
            nrec[c] = ch[c].nrec;
        }
    }
}

// Blocks and records (each k_write lane's run rounded up to whole bursts) of one subsequence, from its checkpoints; a k_write
// lane covers P.write_windows consecutive staging windows.  ckp(ph) = checkpoint word y of window ph.
template <class CkY>
__device__ __forceinline__ void subseq_totals(uint32_t nph, uint32_t ww, CkY cky, uint32_t &sb, uint32_t &sr)
{
    sb = sr = 0;
    for (uint32_t ph = 0; ph < nph; ph += ww) {
        uint32_t r = 0;
        for (uint32_t w = 0; w < ww; w++) {
            const uint32_t c = cky(ph + w);
            sb += (c >> 9) & 2047u;
            r += c >> 20;
        }
        sr += (r + (HJD_REC_RUN - 1u)) & ~(HJD_REC_RUN - 1u);
    }
}

// Synchronisation, MODE 0 (k_sync).
// Round 0: every chain decodes its subsequence from a guess (a code boundary at its first bit, z = 0, MCU phase 0) and leaves,
//   next to the end state, a CHECKPOINT at the end of each staging window (P.ckpt: decoder state there + blocks / records
//   decoded inside that window).
// Refinement rounds: a subsequence whose predecessor's end state differs from the start state it was decoded from becomes a
//   WORK ITEM; the items are compacted (__ballot + prefix over the waves' chains) and dealt to the lanes — one per lane while
//   they fit, HJD_SYNC_C per lane in the dense first round — and every wave that holds items walks them window by window on its
//   own (no barrier inside a round):
//   * an item whose state at a window's end EQUALS the checkpoint of the subsequence's previous decode has fallen into step
//     with it — same bit position, same zig-zag index, same block-in-MCU — so the rest of that decode stands: its end state is
//     kept, its totals are corrected by the difference over the windows walked, and the item is done.
//   * an item that reaches the end of its subsequence with a NEW end state walks on into the next subsequence unless that
//     one is an item of this round itself: a state change runs down the chain inside one round instead of one
//     subsequence per round.
// MODE 1 / 2 (repair) use the same machinery on the records and checkpoints k_sync left (no round 0, no warm-up slots).
// NS_ = slots of the workgroup: HJD_SYNC_WG, or HJD_SYNC_WG_SMALL for the images whose whole scan fits one small workgroup
// (k_sync_small: a thumbnail has a dozen subsequences; in a workgroup of 512 lanes it paid for 32 KB of tables, eight waves at
// every barrier and a slot on a CU's two — 8192 images of 64 x 64, 3.8 M images/s, most of it this kernel).
template <int MODE, uint32_t NS_, bool SIBS = false>
__device__ __forceinline__ void sync_body(const HjdParams &P)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    constexpr bool SMALL = NS_ != HJD_SYNC_WG;
    constexpr uint32_t NS = NS_;              // slots
    constexpr uint32_t C = HJD_SYNC_C;        // chains per lane
    constexpr uint32_t T = NS / C;            // lanes
    constexpr uint32_t NVW = NS / 64u;        // "virtual waves": 64 consecutive slots = chain c of wave wv, vw = wv * C + c
    static_assert(NS <= 1024, "work items carry their slot in 10 bits");
    static_assert(NS % (64 * C) == 0 && NVW <= 16, "whole waves; the per-virtual-wave scratch holds 16");

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    static_assert(MODE == 0 || !SMALL, "the repair kernels have one size");
    // Sibling workgroups (small batches only, HjdParams::tk_n): the grid holds tk_n + 1 workgroups per synchronisation workgroup,
    // the siblings first.  A sibling decodes round 0 over again from the same guess with another block-in-MCU label and leaves its
    // checkpoints in a slab of its own; see the refinement rounds.
    static_assert(!SIBS || (MODE == 0 && !SMALL), "siblings: k_sync_sib only");
    const uint32_t TK = SIBS ? P.tk_n : 0u; // (a kernel of its own, k_sync_sib: with the siblings' code merely present k_sync took 5 % longer)
    const uint32_t sib = TK ? TK - blockIdx.x / P.tk_nwg : 0u; // 0 = the workgroup proper
    const uint32_t bix = TK ? blockIdx.x % P.tk_nwg : blockIdx.x;
    const uint32_t bid = bix + (MODE == 0 ? (SMALL ? P.off_syncwg_small : P.off_syncwg) : P.off_repwg);
    const int ii = (int)(MODE == 0 ? (SMALL ? P.map_syncwg_small[bid] : P.map_syncwg[bid]) : P.map_repwg[bid]);
    static_assert(HJD_SYNC_WG % HJD_WRITE_WG == 0, "a repair workgroup covers whole final-pass workgroups");
    const HjdImage &im = P.img[ii];
    const uint32_t wg = bid - (MODE == 0 ? im.syncwg0 : im.repwg0);
    const uint32_t K = MODE == 0 ? P.sync_overlap : 0u, Q = NS - K;
    const uint32_t own0 = wg * Q, own1 = own0 + Q;
    const uint32_t row0 = own0 >= K ? own0 - K : 0u;

    const uint32_t S = P.subseq_bits;
    const uint32_t total_bits = P.st[ii].ustream_len * 8u;
    const uint32_t nsub = (total_bits + S - 1) / S;
    const uint32_t lim = min(own1, nsub); // subsequences [row0, lim) are this workgroup's slots
    const uint32_t nph = (S >> 5) / P.phase_words;
    const uint32_t nwarm = own0 - row0;

    // this lane's slots
    uint32_t sl[C], row[C];
    bool valid[C], owned[C];
#pragma unroll
    for (uint32_t c = 0; c < C; c++) {
        sl[c] = (wv * C + c) * 64u + lane;
        row[c] = row0 + sl[c];
        valid[c] = row[c] < lim;
        owned[c] = valid[c] && row[c] >= own0;
    }

    // LDS: tables | per-slot state | scratch | block tables | staging windows
    uint32_t *s_lut = s_dyn;
    uint32_t *s_sec = s_dyn + P.lds_sec_word_s;
    uint32_t *s_ep = s_dyn + P.lds_state_word_s; // end state: bit position ...
    uint32_t *s_up = s_ep + NS;                  // the start state that end state was computed from: bit position ...
    uint32_t *s_cnt = s_up + NS;                 // blocks | records << 16 inside the subsequence
    uint32_t *s_mk0 = s_cnt + NS;                // restart streams: index of the first marker past the slot's first bit
    uint16_t *s_ezm = reinterpret_cast<uint16_t *>(s_mk0 + NS); // ... end state z | m << 8
    uint16_t *s_uzm = s_ezm + NS;                               // ... start state z | m << 8
    uint8_t *s_mark = reinterpret_cast<uint8_t *>(s_mk0 + 2 * NS); // last round (+1, <= 96) in which the slot was a work item
    uint32_t *s_wc = s_mk0 + 2 * NS + NS / 4;    // [0..15] work items per virtual wave, [16..31] the virtual waves' run heads, [32] scratch
    uint2 *s_mt = reinterpret_cast<uint2 *>(s_wc + 48);  // 8 entries: block table of the one-chain loop (restart streams)
    uint4 *s_mt4 = reinterpret_cast<uint4 *>(s_wc + 64); // 8 entries: block table of sync_window
    uint32_t *s_rows = s_wc + 96;
    // The work list — start bit position; z | m << 6 | slot << 9 — lives in the first 2 NS words of the staging windows: the
    // list is read into registers, behind a barrier, before a round's first window is staged.
    uint32_t *s_lp = s_rows;
    uint32_t *s_lst = s_rows + NS;
    // k_sync_sib only (its LDS is HJD_SIB_LDS_EXTRA bytes larger), behind the staging windows: per slot the END state of every
    // sibling's trajectory ([t] for slab t), packed into 16 bits (position past the slot's last bit, 5 bits | z << 5 | m << 11 |
    // 1 << 15), and the slab a slot has adopted in this round (0: none).  With these a walk adopts a slot by LDS accesses alone — with
    // the checkpoints in global memory every adopted slot was two or three dependent round trips, 10 us a slot for the one lane of
    // a chain round — and the checkpoints are copied by all lanes at once after the round (materialise).
    constexpr uint32_t NTRAJ = HJD_MAX_SIBLINGS + 1u; // 12
    uint16_t *s_exit = reinterpret_cast<uint16_t *>(s_rows + (NS / 64u) * (P.phase_words + 2u) * 64u);
    uint8_t *s_asg = reinterpret_cast<uint8_t *>(s_exit + NS * NTRAJ);
    static_assert(HJD_SIB_LDS_EXTRA >= HJD_SYNC_WG * ((HJD_MAX_SIBLINGS + 1u) * 2u + 1u), "k_sync_sib's extra LDS");
    const HjdTableSet *ts = &P.tabs[im.tabset];

    RowEnv E;
    E.P = &P;
    E.im = &im;
    E.L = {reinterpret_cast<const uint16_t *>(s_lut), reinterpret_cast<const uint16_t *>(s_sec), s_mt};
    E.tabY = table_pair(ts->slot_pack, 0, 4u);
    E.tabCb = table_pair(ts->slot_pack, 1, 4u);
    E.tabCr = table_pair(ts->slot_pack, 2, 4u);
    fill_mcu_table(s_mt, im.nb, im.nby, E.tabY, E.tabCb, E.tabCr);
    E.mt4_addr = lds_addr(s_mt4);
    E.sec_addr = lds_addr(s_sec);
    fill_block_table(s_mt4, E.mt4_addr, im.nb, im.nby, lds_addr(s_lut), E.tabY, E.tabCb, E.tabCr, tid);
    E.wbase = reinterpret_cast<const uint32_t *>(P.ustream + im.ustream_off);
    E.s_stage = s_rows + wv * (C * (P.phase_words + 2u) * 64u);
    E.total_bits = total_bits;
    E.nmark = im.restart ? min(P.st[ii].nmarkers, im.mark_cap) : 0u;

    // Checkpoints (HjdParams::ckpt): a subsequence the workgroup OWNS has its entries at their canonical place — the
    // image's [window][subsequence] array, which k_write and the repair kernels read — its warm-up slots' entries live in a
    // scratch area private to the workgroup.
    uint2 *const ck_canon = P.ckpt + (size_t)im.sub0 * nph + row0;       // + window * nsub_cap + slot
    uint2 *const ck_scr = P.ckpt + P.ck_scratch + (size_t)bid * nph * K; // + window * K + slot        (MODE 0, slot < nwarm)
    auto ck = [&](uint32_t ph, uint32_t slot) -> uint2 * {
        return slot >= nwarm ? ck_canon + (size_t)ph * im.nsub_cap + slot : ck_scr + ph * K + slot;
    };

    // sibling t's checkpoints of the subsequences this workgroup owns (same layout as the canonical ones, in slab t)
    auto tkck = [&](uint32_t t, uint32_t ph, uint32_t slot) -> uint2 * {
        return P.tk_ckpt + (size_t)t * P.tk_stride + (size_t)im.sub0 * nph + row0 + (size_t)ph * im.nsub_cap + slot;
    };

    uint32_t ext_p = row0 * S, ext_zm = 0; // start state offered to slot 0 from outside the workgroup
    // HJD_SYNC_PROF (diagnostics, tools/sync_profile.py): shader-clock stamps of this workgroup's stages
    unsigned long long *dbg = (MODE == 0 && !SMALL && sib == 0u && P.dbg) ? P.dbg + (size_t)bid * 32u : nullptr;
    uint32_t dbg_k = 0;
    auto stamp = [&](unsigned long long extra) {
        if (dbg && tid == 0 && dbg_k < 30u) {
            dbg[dbg_k++] = __builtin_readcyclecounter();
            dbg[dbg_k++] = extra;
        }
    };
    stamp(nsub);
    uint32_t ndec_total = 0, nphase_items = 0;
#pragma unroll
    for (uint32_t c = 0; c < C; c++) s_mark[sl[c]] = 0;
    if (tid == 0) s_wc[33] = 0u;

    if (MODE != 0 && wg > 0u && P.rep_flags) {
        // Repair launches (host re-synchronisation): the workgroups of an image form a CHAIN — workgroup w takes slot 0's start state
        // from the last record of workgroup w - 1, so it waits until that one has left its final records (a flag per workgroup
        // holding this launch's stamp; workgroups are dispatched in the order of their numbers, so the one waited for is running or
        // done; bounded spin: a workgroup that gives up goes on with what is there, as before, and the host comes round again).
        // One host round trip then carries a true state across ALL workgroup borders of an image — a uniform region spanning several
        // synchronisation workgroups (an object on a black background under tables optimised for it: round 4 needed three round
        // trips, one border each, 3.5 ms wall) — instead of one border per round trip.
        if (tid == 0) {
            const uint32_t want = (P.epoch << 2) | (uint32_t)MODE;
            for (uint32_t spin = 0; spin < (1u << 20); spin++) {
                if (__hip_atomic_load(&P.rep_flags[bid - 1u], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == want) break;
                __builtin_amdgcn_s_sleep(32);
            }
        }
        __syncthreads();
    }
    if (MODE == 0) {
        load_tables<true>(ts, s_lut, s_sec);
        __syncthreads();
        stamp(0);
        // round 0: every chain its own subsequence, from the guess
        uint32_t p[C], z[C], m[C], nblk[C], nrec[C], mk[C], tb[C], tr[C], phs[C];
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            p[c] = row[c] * S;
            z[c] = 0;
            m[c] = 0;
            tb[c] = tr[c] = 0;
            mk[c] = 0xFFFFFFFFu;
            s_up[sl[c]] = p[c];
            s_uzm[sl[c]] = 0;
        }
        if (im.restart) {
#pragma unroll
            for (uint32_t c = 0; c < C; c++) {
                uint32_t mk0 = 0;
                if (valid[c]) {
                    const uint32_t *mpos = P.mpos + im.mark_off;
                    uint32_t lo = 0, hi = E.nmark; // first marker past the start of this row
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (mpos[mid] <= p[c]) lo = mid + 1;
                        else hi = mid;
                    }
                    mk0 = lo; // kept per subsequence: refinement walks, repair launches and k_write start their marker scan here
                }
                mk[c] = mk0;
                s_mk0[sl[c]] = mk0;
                if (owned[c]) P.submk[im.sub0 + row[c]] = mk0;
            }
        } else {
#pragma unroll
            for (uint32_t c = 0; c < C; c++) s_mk0[sl[c]] = 0u;
        }
        for (uint32_t ph = 0; ph < nph; ph++) {
#pragma unroll
            for (uint32_t c = 0; c < C; c++) phs[c] = ph;
            if (sib != 0u && ph == 0u) {
                // A sibling: the same guess with ANOTHER block-in-MCU label (1, 2, 3: for 4:2:0 the other luma blocks).  What a guessed
                // decode has wrong longest is the block's place in the MCU (it finds the code boundaries within a few hundred
                // bits); four guesses are four draws.  (Forking the labels from the first guess's state after one staging window,
                // of the same table class first, was the first form: one 4K image 0.240 ms of k_sync against 0.218 this way.)
                // Siblings nb .. 2 nb - 1: the same labels, but one symbol INTO the block (z = 1: the DC symbol is behind).  On flat
                // content under tables optimised for it (one-bit codes: every bit is a code boundary) a decode that enters at the
                // subsequence's first bit is then always one of the 2 nb trajectories from its very start (the sweep below).
#pragma unroll
                for (uint32_t c = 0; c < C; c++) {
                    m[c] = sib % im.nb;
                    z[c] = (sib / im.nb) & 1u;
                }
            }
            decode_windows<(int)C>(E, valid, row, phs, p, z, m, nblk, nrec, mk);
            if (sib != 0u) {
#pragma unroll
                for (uint32_t c = 0; c < C; c++)
                    if (owned[c]) *tkck(sib - 1u, ph, sl[c]) = make_uint2(p[c], z[c] | (m[c] << 6) | (nblk[c] << 9) | (nrec[c] << 20));
                continue;
            }
#pragma unroll
            for (uint32_t c = 0; c < C; c++) {
                if (valid[c]) *ck(ph, sl[c]) = make_uint2(p[c], z[c] | (m[c] << 6) | (nblk[c] << 9) | (nrec[c] << 20));
                tb[c] += nblk[c];
                tr[c] += nrec[c];
            }
        }
        if (sib != 0u) { // the checkpoints are out: tell the workgroup proper (release, device scope: it may run on another XCD)
            __threadfence();
            __syncthreads();
            if (tid == 0) __hip_atomic_store(&P.tk_flags[(size_t)bid * TK + (sib - 1u)], P.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            s_ep[sl[c]] = p[c];
            s_ezm[sl[c]] = (uint16_t)(z[c] | (m[c] << 8));
            s_cnt[sl[c]] = tb[c] | (tr[c] << 16);
        }
        ndec_total = lim > row0 ? lim - row0 : 0u;
    } else {
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            const uint32_t g = im.sub0 + row[c];
            uint4 r = make_uint4(row[c] * S, 0u, row[c] * S, 0u);
            uint32_t cn = 0;
            if (valid[c]) {
                r = P.substate[g];
                cn = P.subcnt[g];
            }
            s_mk0[sl[c]] = valid[c] && im.restart ? P.submk[g] : 0u;
            s_ep[sl[c]] = r.x;
            s_ezm[sl[c]] = (uint16_t)r.y;
            s_cnt[sl[c]] = cn;
            s_up[sl[c]] = r.z;
            s_uzm[sl[c]] = (uint16_t)r.w;
            if (sl[c] == 0u) {
                ext_p = r.z;
                ext_zm = r.w;
                if (valid[c] && row[c] > 0) {
                    const uint4 pe = P.substate[g - 1];
                    if (pe.x >= row[c] * S && pe.x < row[c] * S + 32u) { // a record's end always lies inside the next word
                        ext_p = pe.x;
                        ext_zm = pe.y;
                    }
                }
            }
        }
    }
    stamp(2); // this wave's round 0 done
    __syncthreads();
    stamp(3); // everybody's
    // The siblings' checkpoints: waited for a bounded time (they started before this workgroup and do the same work); one that
    // has not reported is simply not used — its checkpoints are hints, never results.
    uint32_t tk_ok = 0u;
    if (TK) {
        if (tid < TK) {
            bool ok = false;
            for (uint32_t spin = 0; spin < 4000u && !ok; spin++) {
                ok = __hip_atomic_load(&P.tk_flags[(size_t)bid * TK + tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == P.epoch;
                if (!ok) __builtin_amdgcn_s_sleep(16);
            }
            if (ok) atomicOr(&s_wc[33], 1u << tid);
        }
        __syncthreads();
        tk_ok = s_wc[33];
        __syncthreads();
    }
    stamp(4);

    // Entry adoption (sibling trajectories only).  A decode that enters owned slot s exactly at its first bit, at a block's start or
    // one symbol into it (z <= 1), with label m, IS sibling z * nb + m's decode of s from the first bit on: that sibling's checkpoints,
    // counts and end state are the slot's, nothing is decoded.  Flat content under optimised tables (one-bit codes) enters nearly
    // every subsequence that way, and no other trajectory ever falls into step with it there — a 4K frame of flat patches took
    // 13-24 ms of single-lane walking.  Used whenever a refinement walk starts on a slot or moves on into the next one; and from
    // the fourth round on a round with many items left works off only the FIRST of them (a chain round, below), so that its walk runs
    // down the workgroup — a few loads per adopted slot, a decode where a slot is entered otherwise — instead of every slot being
    // redone from its predecessor's stale state round after round, the truth moving on by one slot per round.  (A sweep over all
    // slots by one lane BEFORE the rounds did the same for flat content but cost ordinary and standard-table content 0.1-0.5 ms.)
    auto pack_state = [&](uint32_t pos, uint32_t base_bit, uint32_t zz, uint32_t mm) -> uint32_t { // 0 = none (not inside the next word)
        return (pos >= base_bit && pos < base_bit + 32u) ? ((pos - base_bit) | (zz << 5) | (mm << 11) | 0x8000u) : 0u;
    };
    auto adopt_slab = [&](uint32_t s, uint32_t ap, uint32_t az, uint32_t am) -> int { // the slab that holds slot s decoded from this state, or -1
        if (!SIBS || im.restart || s < nwarm) return -1;
        const uint32_t bit_s = (row0 + s) * S;
        if (ap == bit_s && az <= 1u) {
            const uint32_t cls = az * im.nb + am; // 0: the canonical decode's own start state
            return (cls != 0u && cls <= TK && ((tk_ok >> (cls - 1u)) & 1u) && s_exit[s * NTRAJ + cls - 1u] != 0u) ? (int)cls - 1 : -1;
        }
        return -1;
    };
    auto can_adopt = [&](uint32_t s, uint32_t ap, uint32_t az, uint32_t am) -> bool { return adopt_slab(s, ap, az, am) >= 0; };
    auto entry_adopt = [&](uint32_t s, uint32_t &ap, uint32_t &az, uint32_t &am) -> bool {
        const int t = adopt_slab(s, ap, az, am);
        if (t < 0) return false;
        const uint32_t e = s_exit[s * NTRAJ + (uint32_t)t];
        s_up[s] = ap;
        s_uzm[s] = (uint16_t)(az | (am << 8));
        ap = (row0 + s + 1u) * S + (e & 31u);
        az = (e >> 5) & 63u;
        am = (e >> 11) & 7u;
        s_ep[s] = ap;
        s_ezm[s] = (uint16_t)(az | (am << 8));
        s_asg[s] = (uint8_t)(t + 1); // checkpoints and counts follow after the round (materialise)
        return true;
    };
    // every lane: the slot's adopted trajectory becomes its own (checkpoints copied, counts added up)
    auto materialise = [&]() {
        if (!SIBS || TK == 0u) return;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            const uint32_t sm = sl[c], t = s_asg[sm];
            if (t == 0u) continue;
            uint32_t cn = 0;
            for (uint32_t w = 0; w < nph; w++) {
                const uint2 v = *tkck(t - 1u, w, sm);
                *ck(w, sm) = v;
                cn += ((v.y >> 9) & 2047u) | ((v.y >> 20) << 16);
            }
            s_cnt[sm] = cn;
            s_asg[sm] = 0u;
        }
    };
    if (SIBS) { // the siblings' end states (their slabs are complete: tk_ok)
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            const uint32_t sm = sl[c];
            s_asg[sm] = 0u;
            uint2 v[NTRAJ]; // all loads first: one round trip
#pragma unroll
            for (uint32_t t = 0; t < NTRAJ; t++) {
                v[t] = make_uint2(0u, 0u);
                if (t < TK && ((tk_ok >> t) & 1u) && valid[c] && sm >= nwarm) v[t] = *tkck(t, nph - 1u, sm);
            }
#pragma unroll
            for (uint32_t t = 0; t < NTRAJ; t++)
                s_exit[sm * NTRAJ + t] = (uint16_t)((t < TK && ((tk_ok >> t) & 1u) && valid[c] && sm >= nwarm)
                                                        ? pack_state(v[t].x, (row0 + sm + 1u) * S, v[t].y & 63u, (v[t].y >> 6) & 7u) : 0u);
        }
        __syncthreads();
    }
    // Uniform regions.  Every MCU of a uniform image is the same few bits (4:2:0 with the standard tables: 32; with tables
    // optimised for the image 12 or less), so the stream is periodic, and on a periodic stream a guessed decode either falls into
    // step within a few symbols or locks into a wrong cycle for good, depending only on the bit phase it starts at (standard
    // tables, 4:2:0: 12 of the 32 phases never fall into step, all subsequences share ONE phase, white and black frames have an
    // unlucky one).  Then no checkpoint is ever met and a single lane walks the true state through the whole region (a white 4K
    // frame: 58 ms instead of 0.3; optimised tables: 60-130 ms).  But a decode is a function of (the words, the start state):
    // where subsequence j consists of the same words as subsequence j - q (its two look-ahead words included) and is handed the
    // start state of j - d moved d S bits on, d a multiple of q, it has j - d's checkpoints, counts and end state, moved on by
    // the same distance, and nothing needs decoding.  That is exact whatever the states are; it resolves a whole run in one
    // round once the run's first d subsequences are settled and the state has come round after them (end of the d-th = start of
    // the first + d S bits, same z and m: d S is then a multiple of the MCU's length).
    //   runq : the distance q (one per workgroup: the smallest any slot found, 1 for a stream of one repeated word)
    //   elig : this slot's words equal those of the slot runq before it;  head : the nearest slot at or below that is not elig
    //   (the run's first runq slots end there).  Restart streams: their markers are not in the words; left to the ordinary rounds.
#ifndef HJD_CHAIN_FROM
#define HJD_CHAIN_FROM 3u // first refinement round that may be a chain round
#endif
#ifndef HJD_RUN_Q
#define HJD_RUN_Q 32u // distances tried (the odd part of the MCU's bit length, in practice)
#define HJD_RUN_D 96u // longest state period, in subsequences
#endif
    bool elig[C], any_elig = false; // any_elig: some slot of this workgroup continues a run (ordinary content: none)
    bool lastrep = false;
    uint32_t head[C], runq = 0;
#pragma unroll
    for (uint32_t c = 0; c < C; c++) {
        elig[c] = false;
        head[c] = 0;
    }
    if (!im.restart) {
        uint32_t myq[C];
        static_assert(HJD_RUN_Q <= 32, "one bit per distance");
        const uint32_t nwords = (S >> 5) + 2u;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            const bool full = valid[c] && (row[c] + 1u) * S <= total_bits;
            const uint32_t *rw = row_base(E.wbase, row[c], P.log2_words);
            const uint32_t w0 = full ? rw[0] : 0u;
            // distances at which the first word agrees (independent loads: one round trip for all of them) ...
            uint32_t cm = 0;
            if (full) {
                const uint32_t qn = min(sl[c], HJD_RUN_Q);
                for (uint32_t q = 1; q <= qn; q++) cm |= (row_base(E.wbase, row[c] - q, P.log2_words)[0] == w0 ? 1u : 0u) << (q - 1u);
            }
            // ... and of those the smallest at which every word does (ordinary content: none to try)
            myq[c] = 0;
            while (__any(cm != 0u && myq[c] == 0u)) {
                bool t = cm != 0u && myq[c] == 0u;
                const uint32_t q = t ? (uint32_t)__ffs((int)cm) : 1u;
                cm &= cm - 1u;
                const uint32_t *rp = t ? row_base(E.wbase, row[c] - q, P.log2_words) : rw;
                for (uint32_t k = 1; k < nwords && __any(t); k++)
                    if (t) t = rw[(size_t)k * 64u] == rp[(size_t)k * 64u];
                if (t) myq[c] = q;
            }
        }
        if (tid == 0) s_wc[32] = 0xFFFFFFFFu;
        __syncthreads();
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            if (myq[c]) atomicMin(&s_wc[32], myq[c]);
            lastrep = lastrep || (myq[c] != 0u && row[c] + 1u == min(own1, nsub) && own1 < nsub); // the workgroup's last owned slot repeats an earlier one
        }
        __syncthreads();
        runq = s_wc[32];
        unsigned long long nb0[C];
        bool some = false;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            elig[c] = myq[c] != 0u && myq[c] == runq;
            some = some || elig[c];
            nb0[c] = __ballot(!elig[c]);
            const uint32_t vw = wv * C + c;
            if (lane == 0) s_wc[16u + vw] = nb0[c] ? vw * 64u + 63u - (uint32_t)__clzll((long long)nb0[c]) : 0xFFFFFFFFu;
        }
        any_elig = __syncthreads_or(some ? 1 : 0) != 0;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            const uint32_t vw = wv * C + c;
            const unsigned long long below = nb0[c] & (~0ull >> (63u - lane));
            if (below) {
                head[c] = vw * 64u + 63u - (uint32_t)__clzll((long long)below);
            } else {
                for (uint32_t q = vw; q-- > 0u;)
                    if (s_wc[16u + q] != 0xFFFFFFFFu) {
                        head[c] = s_wc[16u + q];
                        break;
                    }
            }
        }
    }

    // A uniform region across a workgroup border (round 5).  The warm-up subsequences give a workgroup the true state at its first
    // owned subsequence because a guessed decode falls into step with the true one — except on a periodic stream (a black or white
    // background), where it may lock into a wrong cycle for good: the run rule above then hands the wrong states down the whole
    // workgroup, k_write's check catches the border and the HOST re-synchronises (round 4: one round trip per run of such an image,
    // 2-3 ms wall for an object on a black background under tables optimised for it).  Where every warm-up subsequence repeats the
    // words of an earlier one (looking back beyond the workgroup's own rows) the workgroup instead WAITS for the workgroup before it
    // — which publishes a flag when its own last subsequence continues a run — and takes slot 0's start state from that
    // workgroup's final record, as a repair launch would.  Workgroups are dispatched in the order of their numbers, so the one waited
    // for is running or done; the spin is bounded and a workgroup that gives up goes on from its guess, as before.  Ordinary content
    // never waits and never publishes (its subsequences differ in their first words).
    bool publish = false;
    if (MODE == 0 && !SMALL && sib == 0u && !im.restart && P.sync_flags) {
        bool uni = false; // this warm-up slot repeats an earlier subsequence
        if (wv == 0u && lane < nwarm && wg > 0u) {
            const uint32_t r = row0 + lane;
            if (r > 0u && (r + 1u) * S <= total_bits) {
                const uint32_t nwords = (S >> 5) + 2u;
                const uint32_t *rw = row_base(E.wbase, r, P.log2_words);
                const uint32_t w0 = rw[0], qn = min(r, HJD_RUN_Q);
                uint32_t cm = 0; // distances at which the first word agrees (independent loads: one round trip; ordinary content: none)
                for (uint32_t q = 1; q <= qn; q++) cm |= (row_base(E.wbase, r - q, P.log2_words)[0] == w0 ? 1u : 0u) << (q - 1u);
                while (cm != 0u && !uni) {
                    const uint32_t q = (uint32_t)__ffs((int)cm);
                    cm &= cm - 1u;
                    const uint32_t *rp = row_base(E.wbase, r - q, P.log2_words);
                    bool same = true;
                    for (uint32_t k = 1; k < nwords && same; k++) same = rw[(size_t)k * 64u] == rp[(size_t)k * 64u];
                    uni = same;
                }
            }
        }
        if (tid == 0) s_wc[34] = 0u, s_wc[35] = 0u;
        __syncthreads();
        if (wv == 0u) {
            const unsigned long long bu = __ballot(uni);
            if (lane == 0 && nwarm != 0u && wg > 0u && (uint32_t)__popcll(bu) == nwarm) s_wc[34] = 1u; // every warm-up slot: wait
        }
        // the last owned slot continues a run: the next workgroup may be waiting for this one's records
        if (lastrep) s_wc[35] = 1u;
        __syncthreads();
        publish = s_wc[35] != 0u;
        if (s_wc[34] != 0u) {
            if (tid == 0) {
                bool ok = false;
                for (uint32_t spin = 0; spin < (1u << 16) && !ok; spin++) { // (40 ms at most)
                    ok = __hip_atomic_load(&P.sync_flags[bid - 1u], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == P.epoch;
                    if (!ok) __builtin_amdgcn_s_sleep(32);
                }
                if (ok) {
                    const uint4 pe = P.substate[im.sub0 + row0 - 1u];
                    if (pe.x >= row0 * S && pe.x < row0 * S + 32u) { // a record's end always lies inside the next word
                        s_wc[36] = pe.x;
                        s_wc[37] = pe.y;
                        s_wc[34] = 2u;
                    }
                }
            }
            __syncthreads();
            if (s_wc[34] == 2u) {
                ext_p = s_wc[36];
                ext_zm = s_wc[37];
            }
        }
        __syncthreads();
    }

    uint32_t rounds = 0;
    bool tables_loaded = MODE == 0;
    for (; rounds < 96u; rounds++) {
        // start state every slot should have: its predecessor's current end state
        uint32_t wp[C], wzm[C], src[C], shift[C], mep[C], mezm[C];
        bool dirty[C], moved[C];
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            const uint32_t s = sl[c];
            wp[c] = s ? s_ep[s - 1] : ext_p;
            wzm[c] = s ? (uint32_t)s_ezm[s - 1] : ext_zm;
            if (row[c] == 0) {
                wp[c] = 0;
                wzm[c] = 0; // the true start of the scan (loadjpg.cpp:1148-1149)
            }
            dirty[c] = valid[c] && (wp[c] != s_up[s] || wzm[c] != (uint32_t)s_uzm[s]);
            moved[c] = false; // this slot takes the record of a settled slot of its run, moved on, instead of being decoded
            src[c] = shift[c] = mep[c] = mezm[c] = 0;
        }
        if (any_elig) {
            __syncthreads(); // everybody has read the work list's place (s_lp doubles as the flags below)
#pragma unroll
            for (uint32_t c = 0; c < C; c++) s_lp[sl[c]] = dirty[c] ? 1u : 0u;
            __syncthreads();
#pragma unroll
            for (uint32_t c = 0; c < C; c++) {
                if (!elig[c]) continue;
                const uint32_t s = sl[c];
                const uint32_t a = head[c] + 1u - runq; // the run's first slot (slots below runq are never elig: a >= 0)
                uint32_t dsel = 0;
                bool settled = true; // slots [a, a + d): their records are not about to change, each starts where the one before ends
                for (uint32_t d = runq; d <= HJD_RUN_D && a + d <= s && settled; d += runq) {
                    for (uint32_t x = a + d - runq; x < a + d; x++) settled = settled && s_lp[x] == 0u;
                    if (settled && s_ep[a + d - 1u] == s_up[a] + d * S && s_ezm[a + d - 1u] == s_uzm[a]) { // the state has come round
                        dsel = d;
                        break;
                    }
                }
                if (dsel) {
                    const uint32_t t = s - a;
                    src[c] = a + t % dsel;
                    shift[c] = (t / dsel) * dsel * S;
                    wp[c] = s_up[src[c]] + shift[c];
                    wzm[c] = s_uzm[src[c]];
                    mep[c] = s_ep[src[c]] + shift[c];
                    mezm[c] = s_ezm[src[c]];
                    moved[c] = wp[c] != s_up[s] || wzm[c] != (uint32_t)s_uzm[s] || mep[c] != s_ep[s] || mezm[c] != (uint32_t)s_ezm[s];
                    dirty[c] = false;
                }
            }
            __syncthreads(); // the flags have been read: s_lp is the work list again
        }
        bool chain_round = false; // this round has ONE item (the first slot entered in a state it was not decoded from) whose walk runs down the workgroup
        if (SIBS && tk_ok != 0u && !im.restart && rounds >= HJD_CHAIN_FROM) { // uniform
            // A chain round?  Three rounds have not settled the workgroup, many slots are still entered in a state they were not
            // decoded from, and most of those at their first bit in a sibling's start state: flat content (see entry_adopt).  Redoing
            // every such slot from its predecessor's stale end moves the truth on by one slot per round.  Instead only the FIRST of
            // them becomes an item; nothing else is marked, so its walk runs down the workgroup — a few loads per adopted slot, a
            // decode where a slot is entered otherwise — until it reaches a slot that was decoded from the state it brings.
            // (Every head of a stretch of such slots as an item: fewer rounds, 10-22 instead of 17-42 on the flat 4K frames, but
            // longer ones — walks on stale states that block each other: 2.9-4.5 ms against 2.1-4.8.)
            bool dany = false, aany = false;
#pragma unroll
            for (uint32_t c = 0; c < C; c++) {
                dany = dany || dirty[c];
                aany = aany || (dirty[c] && can_adopt(sl[c], wp[c], wzm[c] & 63u, wzm[c] >> 8));
            }
            const uint32_t nd = (uint32_t)__syncthreads_count(dany ? 1 : 0), na = (uint32_t)__syncthreads_count(aany ? 1 : 0);
            if (nd >= 8u && na * 2u >= nd) {
                chain_round = true;
#pragma unroll
                for (uint32_t c = 0; c < C; c++) {
                    const unsigned long long bd = __ballot(dirty[c]);
                    if (lane == 0) s_wc[16u + wv * C + c] = bd ? (wv * C + c) * 64u + (uint32_t)__ffsll((long long)bd) - 1u : 0xFFFFFFFFu;
                }
                __syncthreads();
                uint32_t first = 0xFFFFFFFFu;
#pragma unroll
                for (uint32_t q = 0; q < NVW; q++) first = min(first, s_wc[16u + q]);
                __syncthreads();
#pragma unroll
                for (uint32_t c = 0; c < C; c++) dirty[c] = dirty[c] && sl[c] == first;
            }
        }
        unsigned long long bal[C];
        bool any_moved = false;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            bal[c] = __ballot(dirty[c]);
            any_moved = any_moved || moved[c];
        }
        any_moved = __any(any_moved);
        if (lane == 0) {
#pragma unroll
            for (uint32_t c = 0; c < C; c++) s_wc[wv * C + c] = (uint32_t)__popcll(bal[c]) | (any_moved ? 0x10000u : 0u);
        }
        __syncthreads();
        uint32_t base[C], n = 0;
#pragma unroll
        for (uint32_t c = 0; c < C; c++) base[c] = 0;
#pragma unroll
        for (uint32_t q = 0; q < NVW; q++) {
            const uint32_t cnt = s_wc[q];
#pragma unroll
            for (uint32_t c = 0; c < C; c++) base[c] += q < wv * C + c ? cnt & 0xFFFFu : 0u;
            n += cnt;
        }
        if (n == 0) break;
        n &= 0xFFFFu; // bits 16..: some slot is moved this round (another round then checks the chain)
        if (MODE != 0 && !tables_loaded) {
            load_tables<true>(ts, s_lut, s_sec);
            tables_loaded = true;
        }
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            const uint32_t s = sl[c];
            if (moved[c]) {
                for (uint32_t ph = 0; ph < nph; ph++) {
                    const uint2 cc = *ck(ph, src[c]);
                    *ck(ph, s) = make_uint2(cc.x + shift[c], cc.y);
                }
                s_up[s] = wp[c];
                s_uzm[s] = (uint16_t)wzm[c];
                s_ep[s] = mep[c];
                s_ezm[s] = (uint16_t)mezm[c];
                s_cnt[s] = s_cnt[src[c]];
            }
            // Nobody ever walks on INTO a slot that continues a run: a walk that passes the run's first slots with a new state would go
            // on through the whole run, one lane, window after window — the run takes the new records in the next round instead.
            if (elig[c]) s_mark[s] = (uint8_t)((rounds + 1u) | 128u); // bit 7: not an item — a walk may pass through by entry adoption, never by decoding
            if (dirty[c]) {
                const uint32_t t = base[c] + (uint32_t)__popcll(bal[c] & ((1ull << lane) - 1ull));
                s_lp[t] = wp[c];
                s_lst[t] = (wzm[c] & 63u) | ((wzm[c] >> 8) << 6) | (s << 9);
                s_up[s] = wp[c];
                s_uzm[s] = (uint16_t)wzm[c];
                s_mark[s] = (uint8_t)(rounds + 1u);
            }
        }
        __syncthreads();
        stamp(0x100 + n);
        // The round's items.  While they fit, one per lane (a sparse round is latency: every item gets a lane's whole issue rate);
        // the dense rounds C per lane, side by side.
        const bool dense = n > T;
        uint32_t steps_done = 0;
        auto walk = [&](auto nc_tag, auto sib_tag) {
            constexpr int NC = decltype(nc_tag)::value;
            constexpr bool SIB = decltype(sib_tag)::value; // sibling trajectories to look at (a small batch); else their code is not there
            bool live[NC];
            uint32_t slot[NC], ph[NC], p[NC], z[NC], m[NC], cn[NC], co[NC], wmk[NC], wrow[NC], nblk[NC], nrec[NC];
            bool anylive = false;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const uint32_t t = tid + (uint32_t)c * T;
                live[c] = t < n;
                const uint32_t st = live[c] ? s_lst[t] : 0u;
                p[c] = live[c] ? s_lp[t] : 0u;
                slot[c] = st >> 9;
                ph[c] = 0u;
                z[c] = st & 63u;
                m[c] = (st >> 6) & 7u;
                cn[c] = co[c] = 0u; // blocks | records << 16 of this walk / of the previous decode's checkpoints, windows walked so far
                anylive = anylive || live[c];
            }
            __syncthreads(); // the list is in registers: its place is the staging windows' again
#pragma unroll
            for (int c = 0; c < NC; c++) wmk[c] = live[c] ? s_mk0[slot[c]] : 0u; // the walk's marker scan: never behind the state's position, which only grows
            if (!__any(anylive)) return; // waves without work go straight to the barrier
            // Lane c is about to decode slot[c] from (p, z, m) at window 0: while that is a sibling's start state, the slot is the
            // sibling's (entry_adopt) and the lane moves on to the next one as a walk that reached the slot's end would.
            // may a walk that brings (ap, az, am) enter slot nx this round?  Not an item of the round or a slot another walk has
            // taken; a slot that continues a run only where nothing is decoded (it is a sibling's from its first bit)
            auto may_enter = [&](uint32_t nx, uint32_t ap, uint32_t az, uint32_t am) -> bool {
                const uint32_t mk8 = s_mark[nx];
                if ((mk8 & 127u) != rounds + 1u) return true;
                return (mk8 & 128u) != 0u && SIB && can_adopt(nx, ap, az, am);
            };
            // Chain rounds (round 5): the one walk of the round does not END at a slot that was decoded from the very state it brings —
            // that slot's records stand, the walk passes over it with its end state and goes on behind it.  (Round 4 ended the walk
            // there and started the next round at the next slot entered in a wrong state: 17-42 rounds of a few slots each on the flat
            // 4K frames, 70 us a round whatever it held.)
            auto next_slot = [&](uint32_t nx, uint32_t &ap, uint32_t &az, uint32_t &am) -> uint32_t {
                if (SIB && chain_round) {
                    while (nx < NS && row0 + nx < lim && s_mark[nx] != (uint8_t)(rounds + 1u) && s_up[nx] == ap && (uint32_t)s_uzm[nx] == (az | (am << 8))) {
                        ap = s_ep[nx];
                        const uint32_t e = s_ezm[nx];
                        az = e & 0xFFu;
                        am = e >> 8;
                        nx++;
                    }
                }
                return nx;
            };
            auto move_on = [&](int c) {
                while (live[c]) {
                    uint32_t ap = p[c], az = z[c], am = m[c];
                    if (!entry_adopt(slot[c], ap, az, am)) break;
                    const uint32_t nx = next_slot(slot[c] + 1u, ap, az, am);
                    p[c] = ap, z[c] = az, m[c] = am;
                    const uint32_t ezm = az | (am << 8);
                    live[c] = nx < NS && row0 + nx < lim && may_enter(nx, ap, az, am) && (s_up[nx] != ap || (uint32_t)s_uzm[nx] != ezm);
                    if (live[c]) {
                        s_mark[nx] = (uint8_t)(rounds + 1u);
                        s_up[nx] = ap;
                        s_uzm[nx] = (uint16_t)ezm;
                        slot[c] = nx;
                    }
                }
            };
            if (SIB) {
#pragma unroll
                for (int c = 0; c < NC; c++) move_on(c); // the items themselves
                anylive = false;
#pragma unroll
                for (int c = 0; c < NC; c++) anylive = anylive || live[c];
            }
            for (uint32_t step = 0; __any(anylive) && step < 65536u; step++) {
                uint2 old[NC], sbl[NC][HJD_MAX_SIBLINGS];
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    old[c] = make_uint2(0xFFFFFFFFu, 0u);
                    if (live[c]) old[c] = *ck(ph[c], slot[c]); // arrives while the window is decoded
                    wrow[c] = row0 + slot[c];
#pragma unroll
                    for (uint32_t t = 0; t < HJD_MAX_SIBLINGS; t++) { // the siblings' states at this window's end (owned subsequences)
                        sbl[c][t] = make_uint2(0xFFFFFFFFu, 0u);
                        if (SIB && live[c] && ((tk_ok >> t) & 1u) && slot[c] >= nwarm) sbl[c][t] = *tkck(t, ph[c], slot[c]);
                    }
                }
                decode_windows<NC>(E, live, wrow, ph, p, z, m, nblk, nrec, wmk);
                steps_done++;
                anylive = false;
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    if (!live[c]) continue;
                    nphase_items++;
                    cn[c] += nblk[c] | (nrec[c] << 16);
                    const uint32_t st9 = z[c] | (m[c] << 6);
                    const bool met = p[c] == old[c].x && st9 == (old[c].y & 511u);
                    co[c] += ((old[c].y >> 9) & 2047u) | ((old[c].y >> 20) << 16);
                    *ck(ph[c], slot[c]) = make_uint2(p[c], st9 | (nblk[c] << 9) | (nrec[c] << 20));
                    int sibmet = -1; // in step with a sibling's decode instead?
                    if (SIB && !met) {
#pragma unroll
                        for (int t = (int)HJD_MAX_SIBLINGS - 1; t >= 0; t--)
                            if (p[c] == sbl[c][t].x && st9 == (sbl[c][t].y & 511u)) sibmet = t;
                    }
                    if (met) {
                        // in step with the previous decode from here on: its end state stands, its totals change by the
                        // difference over the windows walked (field-wise: no field of the sum can borrow)
                        s_cnt[slot[c]] = s_cnt[slot[c]] + cn[c] - co[c];
                        live[c] = false;
                    } else if (sibmet >= 0 && ph[c] + 1u < nph) {
                        // The rest of the subsequence is the sibling's: its checkpoints become the subsequence's own, its end state
                        // the subsequence's.  If that is not the state the next subsequence was decoded from, the walk goes on there.
                        uint2 last = make_uint2(0u, 0u);
                        for (uint32_t w = ph[c] + 1u; w < nph; w++) {
                            last = *tkck((uint32_t)sibmet, w, slot[c]);
                            *ck(w, slot[c]) = last;
                            cn[c] += ((last.y >> 9) & 2047u) | ((last.y >> 20) << 16);
                        }
                        p[c] = last.x;
                        z[c] = last.y & 63u;
                        m[c] = (last.y >> 6) & 7u;
                        const uint32_t ezm = z[c] | (m[c] << 8);
                        s_ep[slot[c]] = p[c];
                        s_ezm[slot[c]] = (uint16_t)ezm;
                        s_cnt[slot[c]] = cn[c];
                        const uint32_t nx = next_slot(slot[c] + 1u, p[c], z[c], m[c]);
                        const uint32_t ezm2 = z[c] | (m[c] << 8);
                        live[c] = nx < NS && row0 + nx < lim && may_enter(nx, p[c], z[c], m[c]) && (s_up[nx] != p[c] || (uint32_t)s_uzm[nx] != ezm2);
                        if (live[c]) {
                            s_mark[nx] = (uint8_t)(rounds + 1u);
                            s_up[nx] = p[c];
                            s_uzm[nx] = (uint16_t)ezm2;
                            slot[c] = nx;
                            ph[c] = 0u;
                            cn[c] = 0u;
                            co[c] = 0u;
                            if (SIB) move_on(c);
                        }
                    } else if (ph[c] + 1u < nph) {
                        ph[c]++;
                    } else { // end of the subsequence with a state of its own
                        s_ep[slot[c]] = p[c];
                        s_ezm[slot[c]] = (uint16_t)(z[c] | (m[c] << 8));
                        s_cnt[slot[c]] = cn[c];
                        const uint32_t nx = next_slot(slot[c] + 1u, p[c], z[c], m[c]);
                        live[c] = nx < NS && row0 + nx < lim && may_enter(nx, p[c], z[c], m[c]); // walk on: nobody else works on nx this round
                        if (live[c]) {
                            s_mark[nx] = (uint8_t)(rounds + 1u);
                            s_up[nx] = p[c];
                            s_uzm[nx] = (uint16_t)(z[c] | (m[c] << 8));
                            slot[c] = nx;
                            ph[c] = 0u;
                            cn[c] = 0u;
                            co[c] = 0u;
                            if (SIB) move_on(c);
                        }
                    }
                    anylive = anylive || live[c];
                }
            }
        };
        if (SIBS && tk_ok) {
            if (dense) walk(std::integral_constant<int, (int)C>{}, std::true_type{});
            else walk(std::integral_constant<int, 1>{}, std::true_type{});
        } else {
            if (dense) walk(std::integral_constant<int, (int)C>{}, std::false_type{});
            else walk(std::integral_constant<int, 1>{}, std::false_type{});
        }
        stamp(0x10000 + steps_done); // wave 0's windows
        __syncthreads();
        if (SIBS) {
            materialise();
            __syncthreads();
        }
        stamp(0x20000);
    }
    {   // diagnostics: windows walked by all lanes -> subsequence decodes
        __syncthreads();
        if (tid == 0) s_wc[32] = 0u;
        __syncthreads();
        if (nphase_items) atomicAdd(&s_wc[32], nphase_items);
        __syncthreads();
        ndec_total += s_wc[32] / nph;
        __syncthreads();
    }

    if (MODE == 0) {
        // Totals of the final-pass workgroups: k_write runs one lane per staging WINDOW and places its outputs by prefix
        // sums over the windows' block / record counts; the sums over everything BEFORE a final-pass workgroup come from
        // per-workgroup totals (P.wgsum, zeroed by k_unstuff_count).  A final-pass workgroup covers G
        // subsequences, which a synchronisation workgroup's range [own0, own1) cuts anywhere: LDS atomics per group, then one
        // global atomic per group and counter.  A stale workgroup border is still caught: k_write
        // checks the induction step for every subsequence and the host re-synchronises.
        const uint32_t G = (im.write_small ? HJD_WRITE_WG_SMALL : HJD_WRITE_WG) / (nph / P.write_windows); // subsequences per final-pass workgroup (powers of two)
        const uint32_t wq0 = own0 / G;
        __syncthreads(); // everyone is done with the work list
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            s_lp[sl[c]] = 0u;
            s_lst[sl[c]] = 0u;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t c = 0; c < C; c++)
            if (owned[c]) {
                uint32_t sb, sr;
                const uint32_t s = sl[c];
                subseq_totals(nph, P.write_windows, [&](uint32_t ph) { return ck(ph, s)->y; }, sb, sr);
                atomicAdd(&s_lp[row[c] / G - wq0], sb);
                atomicAdd(&s_lst[row[c] / G - wq0], sr);
            }
        __syncthreads();
        for (uint32_t t = tid; t < NS; t += T) {
            const uint32_t wq = wq0 + t;
            if (t <= (own1 - 1u) / G - wq0 && wq < im.nwritewg && (s_lp[t] | s_lst[t])) {
                atomicAdd(&P.wgsum[im.wgsum0 + wq].x, s_lp[t]);
                atomicAdd(&P.wgsum[im.wgsum0 + wq].y, s_lst[t]);
            }
        }
        if (wg == 0 && tid == 0 && nsub == 0) atomicOr(&P.st[ii].flags, HJD_FLAG_SHORT); // a scan without a single data byte
    }
    if (MODE == 2) {
        // The same totals after a repair (host re-synchronisation): written, not added — a repair workgroup holds whole groups.
        const uint32_t G = (im.write_small ? HJD_WRITE_WG_SMALL : HJD_WRITE_WG) / (nph / P.write_windows);
        __syncthreads(); // everyone is done with the work list
#pragma unroll
        for (uint32_t c = 0; c < C; c++) {
            uint32_t sb = 0, sr = 0;
            const uint32_t s = sl[c];
            if (valid[c]) subseq_totals(nph, P.write_windows, [&](uint32_t ph) { return ck(ph, s)->y; }, sb, sr);
            s_lp[s] = sb;
            s_lst[s] = sr;
        }
        __syncthreads();
        for (uint32_t t = tid; t * G < NS; t += T) { // one lane per final-pass workgroup of this repair workgroup
            uint32_t tb = 0, tr = 0;
            for (uint32_t q = 0; q < G; q++) {
                tb += s_lp[t * G + q];
                tr += s_lst[t * G + q];
            }
            const uint32_t wq = wg * (NS / G) + t;
            if (wq < im.nwritewg) P.wgsum[im.wgsum0 + wq] = make_uint2(tb, tr);
        }
        if (wg == 0 && tid == 0 && nsub == 0) atomicOr(&P.st[ii].flags, HJD_FLAG_SHORT); // a scan without a single data byte
    }
    if (!(MODE != 0 && rounds == 0)) { // (repair with nothing to repair: the records stay as they are)
#pragma unroll
        for (uint32_t c = 0; c < C; c++)
            if (owned[c]) {
                const uint32_t s = sl[c], g = im.sub0 + row[c];
                P.substate[g] = make_uint4(s_ep[s], (uint32_t)s_ezm[s], s_up[s], (uint32_t)s_uzm[s]);
                P.subcnt[g] = s_cnt[s];
            }
        if (tid == 0) { // diagnostics
            atomicAdd(&P.st[ii].sync_decodes, ndec_total);
            atomicAdd(&P.st[ii].sync_iters, rounds);
            atomicMax(&P.st[ii].sync_iters_max, rounds);
        }
    }
    if (MODE == 0 && publish) { // the records of a workgroup whose last subsequences are a uniform run: the next one may be waiting for them
        __threadfence();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&P.sync_flags[bid], P.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (MODE != 0 && P.rep_flags) { // this workgroup's records are final for this launch: the next workgroup of the chain may go on
        __threadfence();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&P.rep_flags[bid], (P.epoch << 2) | (uint32_t)MODE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(HJD_SYNC_THREADS) void k_sync(HjdParams P) { sync_body<0, HJD_SYNC_WG>(P); }
__global__ __launch_bounds__(HJD_SYNC_THREADS) void k_sync_sib(HjdParams P) { sync_body<0, HJD_SYNC_WG, true>(P); } // small batches: with sibling trajectories
__global__ __launch_bounds__(HJD_SYNC_WG_SMALL / HJD_SYNC_C) void k_sync_small(HjdParams P) { sync_body<0, HJD_SYNC_WG_SMALL>(P); }

// the repair pass is its own kernel symbol so that profiles tell the two apart
__global__ __launch_bounds__(HJD_SYNC_THREADS) void k_repair(HjdParams P) { sync_body<1, HJD_SYNC_WG>(P); }
__global__ __launch_bounds__(HJD_SYNC_THREADS) void k_repair_totals(HjdParams P) { sync_body<2, HJD_SYNC_WG>(P); }

// k_sync_seq — thumbnails.  A scan of a few kilobytes is a dozen subsequences: with a synchronisation workgroup of its own (even the
// one-wave kind) an image paid for 32 KB of tables, the rounds' barriers and a place on a CU for a few lanes' work — 8192 images of
// 64 x 64: 0.6 ms.  Here ONE LANE decodes ONE image from its first bit to its last, in the synchronisation form, no guessing and
// no rounds; the 64 images of a workgroup share their tables (the host groups images by table set and sampling) and the loop of
// sync_window does not care that its lanes work on different images.  Leaves what k_sync leaves: checkpoints, subsequence
// records, the final-pass workgroups' totals.  The time is that of the longest image of a wave (12 kbit: 0.2 ms).
__global__ __launch_bounds__(64) void k_sync_seq(HjdParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    const uint32_t tid = threadIdx.x;
    const uint32_t bid = blockIdx.x + P.off_seqwg;
    const uint32_t *list = P.seq_list + (size_t)bid * 64u;
    const uint32_t i0 = list[0]; // never empty: the group's first image gives the tables and the sampling
    const uint32_t iid = list[tid];
    const bool have = iid != 0xFFFFFFFFu;
    const HjdImage &im0 = P.img[i0];
    const HjdImage &im = P.img[have ? iid : i0];
    const HjdTableSet *ts = &P.tabs[im0.tabset];

    uint32_t *s_lut = s_dyn;
    uint32_t *s_sec = s_dyn + P.lds_sec_word_s;
    uint32_t *s_w = s_dyn + P.lds_state_word_s;
    uint2 *s_mt = reinterpret_cast<uint2 *>(s_w);        // 8 entries (unused here: restart streams are not sent this way)
    uint4 *s_mt4 = reinterpret_cast<uint4 *>(s_w + 16);  // 8 entries
    uint32_t *s_rows = s_w + 48;                         // one staging window per lane

    RowEnv E;
    E.P = &P;
    E.im = &im;
    E.L = {reinterpret_cast<const uint16_t *>(s_lut), reinterpret_cast<const uint16_t *>(s_sec), s_mt};
    E.tabY = table_pair(ts->slot_pack, 0, 4u);
    E.tabCb = table_pair(ts->slot_pack, 1, 4u);
    E.tabCr = table_pair(ts->slot_pack, 2, 4u);
    E.mt4_addr = lds_addr(s_mt4);
    E.sec_addr = lds_addr(s_sec);
    fill_block_table(s_mt4, E.mt4_addr, im0.nb, im0.nby, lds_addr(s_lut), E.tabY, E.tabCb, E.tabCr, tid);
    E.wbase = reinterpret_cast<const uint32_t *>(P.ustream + im.ustream_off);
    E.s_stage = s_rows;
    E.total_bits = have ? P.st[iid].ustream_len * 8u : 0u;
    E.nmark = 0u;
    load_tables<true>(ts, s_lut, s_sec);
    __syncthreads();

    const uint32_t S = P.subseq_bits, nph = (S >> 5) / P.phase_words;
    const uint32_t nsub = (E.total_bits + S - 1u) / S;
    const uint32_t G = (im.write_small ? HJD_WRITE_WG_SMALL : HJD_WRITE_WG) / (nph / P.write_windows); // subsequences per final-pass workgroup
    uint2 *const ckc = P.ckpt + (size_t)im.sub0 * nph; // + window * nsub_cap + subsequence
    uint32_t p[1] = {0u}, z[1] = {0u}, m[1] = {0u}, nblk[1], nrec[1], mk[1] = {0u}, row[1] = {0u}, ph[1] = {0u};
    uint32_t up = 0u, uzm = 0u;      // the state the current subsequence starts from
    uint32_t sb = 0u, sr = 0u, rr = 0u, tb = 0u, tr = 0u; // blocks / padded records of the current final-pass group, records of the current
                                     // k_write lane (write_windows windows), blocks / records of the subsequence
    bool live[1] = {have && nsub != 0u};
    while (__any(live[0])) {
        decode_windows<1>(E, live, row, ph, p, z, m, nblk, nrec, mk);
        if (live[0]) {
            ckc[(size_t)ph[0] * im.nsub_cap + row[0]] = make_uint2(p[0], z[0] | (m[0] << 6) | (nblk[0] << 9) | (nrec[0] << 20));
            tb += nblk[0];
            tr += nrec[0];
            sb += nblk[0];
            rr += nrec[0];
            if (((ph[0] + 1u) & (P.write_windows - 1u)) == 0u) { // a k_write lane's span ends: its run of records is padded to whole bursts
                sr += (rr + (HJD_REC_RUN - 1u)) & ~(HJD_REC_RUN - 1u);
                rr = 0u;
            }
            if (++ph[0] == nph) { // the subsequence is complete
                const uint32_t g = im.sub0 + row[0], ezm = z[0] | (m[0] << 8);
                P.substate[g] = make_uint4(p[0], ezm, up, uzm);
                P.subcnt[g] = tb | (tr << 16);
                up = p[0];
                uzm = ezm;
                tb = tr = 0u;
                ph[0] = 0u;
                row[0]++;
                if ((row[0] & (G - 1u)) == 0u || row[0] == nsub) { // a final-pass workgroup's subsequences are complete
                    const uint32_t wq = (row[0] - 1u) / G;
                    if (wq < im.nwritewg) P.wgsum[im.wgsum0 + wq] = make_uint2(sb, sr); // (zeroed by k_unstuff_count; nobody else adds here)
                    sb = sr = 0u;
                }
                live[0] = row[0] < nsub;
            }
        }
    }
    if (have) {
        HjdImageState &st = P.st[iid];
        st.sync_decodes = nsub;
        if (nsub == 0u) atomicOr(&st.flags, HJD_FLAG_SHORT); // a scan without a single data byte
    }
}

// ---- stage 3: final entropy decode, coefficients out ------------------------------------------

// One lane per span of P.write_windows staging windows of a subsequence: k_sync's checkpoints hold the decoder state at every
// window's end and the blocks / records decoded inside every window, so the final pass can start anywhere a window starts.
// (Round 2 ran one lane per 2048-bit subsequence: 1.1 resident waves per SIMD, every lane a chain of ~430 symbol steps.)
// A workgroup covers G = HJD_WRITE_WG / (spans per subsequence) whole subsequences:
// their words are loaded once, coalesced, into LDS ([word][row], rows + 1 apart), and every lane decodes its span from there.
// WGSZ lanes: WGSZ, or WGSZ_SMALL (one wave) for the images with a small synchronisation workgroup (k_write_small:
// a thumbnail is a few dozen staging windows; ENTCAP = block entries collected in LDS).
template <uint32_t WGSZ, uint32_t ENTCAP>
__device__ __forceinline__ void write_body(const HjdParams &P)
{
    constexpr bool SMALL = WGSZ != HJD_WRITE_WG;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t bid = blockIdx.x + (SMALL ? P.off_writewg_small : P.off_writewg);
    const int ii = (int)(SMALL ? P.map_writewg_small[bid] : P.map_writewg[bid]);
    const HjdImage &im = P.img[ii];
    const uint32_t wg = bid - im.writewg0;

    uint16_t *s_lut = reinterpret_cast<uint16_t *>(s_dyn);
    uint16_t *s_sec = reinterpret_cast<uint16_t *>(s_dyn + P.lds_sec_word);
    uint32_t *s_w = s_dyn + P.lds_state_word;
    uint8_t *s_slot = reinterpret_cast<uint8_t *>(s_w + 32);
    uint2 *s_mt = reinterpret_cast<uint2 *>(s_w + 48);
    uint4 *s_mt4 = reinterpret_cast<uint4 *>(s_w + 64); // block table of the hand-written loop (fill_block_table): six entries
    uint32_t *s_ring = s_w + 96;                        // [HJD_REC_RUN][WGSZ]: the lanes' bursts in the making
    uint32_t *s_ent = s_ring + HJD_REC_RUN * WGSZ; // [ENTCAP]: block entries in the making
    uint32_t *s_rows = s_ent + ENTCAP;
    const HjdTableSet *ts = &P.tabs[im.tabset];

    const uint32_t S = P.subseq_bits, wps = S >> 5, nph = wps / P.phase_words;
    const uint32_t WWN = P.write_windows, nu = nph / WWN; // windows per lane, lanes per subsequence
    const uint32_t lnu = (uint32_t)__ffs((int)nu) - 1u;    // nu is a power of two
    const uint32_t G = WGSZ >> lnu, stride = G + 1u; // subsequences of this workgroup; words between a row's words in LDS
    const uint32_t total_bits = P.st[ii].ustream_len * 8u;
    const uint32_t nsub = (total_bits + S - 1) / S;
    if (wg * WGSZ >= nsub * nu) return; // capacity-based grid: nothing in this workgroup
    load_tables<false>(ts, s_lut, s_sec);
    if (tid < 64) s_slot[tid] = c_zz_to_slot[tid];
    for (uint32_t k = tid; k < ENTCAP; k += WGSZ) s_ent[k] = 0xFFFFFFFFu;
    fill_mcu_table(s_mt, im.nb, im.nby, table_pair(ts->slot_pack, 0, 2u), table_pair(ts->slot_pack, 1, 2u), table_pair(ts->slot_pack, 2, 2u));
    fill_block_table(s_mt4, lds_addr(s_mt4), im.nb, im.nby, lds_addr(s_lut), table_pair(ts->slot_pack, 0, 2u), table_pair(ts->slot_pack, 1, 2u),
                     table_pair(ts->slot_pack, 2, 2u), tid);
    {   // the workgroup's rows: word k of row i0 + r -> s_rows[k * stride + r] (rows past the scan's end stay unread)
        const uint32_t i0 = wg * G, nrows = min(G, nsub - i0), lw = P.log2_words;
        const uint32_t *wbase = reinterpret_cast<const uint32_t *>(P.ustream + im.ustream_off);
        const uint32_t lg = (uint32_t)__ffs((int)G) - 1u; // G is a power of two
        for (uint32_t idx = tid; idx < G * (wps + 2u); idx += WGSZ) {
            const uint32_t k = idx >> lg, r = idx & (G - 1u);
            if (r < nrows) s_rows[k * stride + r] = row_base(wbase, i0 + r, lw)[(size_t)k * 64u]; // (non-temporal loads here: no difference)
        }
    }

    // lane tid <-> span u: subsequence u / nu, windows [(u % nu) * WWN, + WWN)
    const uint2 *ckc = P.ckpt + (size_t)im.sub0 * nph; // [window][subsequence] of this image
    const uint32_t u = wg * WGSZ + tid;
    const uint32_t i = u >> lnu, ph = (u & (nu - 1u)) * WWN;
    const bool valid = i < nsub;
    const uint32_t g = im.sub0 + i;
    uint2 mine = make_uint2(0u, 0u); // x, y & 511: state at the end of the span
    uint32_t my_blocks = 0, my_recs = 0;
    if (valid)
        for (uint32_t q = 0; q < WWN; q++) {
            mine = ckc[(size_t)(ph + q) * im.nsub_cap + i];
            my_blocks += (mine.y >> 9) & 2047u;
            my_recs += mine.y >> 20;
        }
    // where the spans' blocks and records go: totals of the workgroups before this one + scan inside
    uint32_t pb = 0, pr = 0;
    for (uint32_t k = tid; k < wg; k += WGSZ) {
        const uint2 t = P.wgsum[im.wgsum0 + k];
        pb += t.x;
        pr += t.y;
    }
    // (two barriers for the four of them: four scans one after the other, three barriers each, were a good part of a lane's
    // prologue; s_w[0..15] and s_w[16..31] are the two helpers' scratch)
    static_assert(WGSZ <= 512, "the helpers' scratch holds eight waves");
    uint32_t base_b = pb, base_r = pr, tot_b, tot_r;
    block_sum2(base_b, base_r, s_w);
    uint32_t blk = my_blocks, nrec = (my_recs + (HJD_REC_RUN - 1u)) & ~(HJD_REC_RUN - 1u);
    block_excl_scan2(blk, nrec, s_w + 16, &tot_b, &tot_r);
    blk += base_b;
    nrec += base_r;
    (void)tot_r;
    uint32_t flags = 0;
    if (tid == 0 && (wg + 1u) * WGSZ >= nsub * nu) { // the workgroup that holds the last span
        P.st[ii].total_blocks = base_b + tot_b;
        if (base_b + tot_b < im.nblocks) flags |= HJD_FLAG_SHORT;
    }
    __syncthreads(); // tables, rows and the prefix arrays are in LDS
    // (Dealing the spans to the lanes in descending order of their length in steps — known from the checkpoints — was tried:
    // 1 % fewer instructions.  Neighbouring spans are of like length already.)

    // start state: the checkpoint at the end of the window before; at a subsequence's first window the induction step
    // is checked — the subsequence's record must have been computed from its predecessor's end state
    uint32_t p = 0, z = 0, m = 0;
    bool go = valid;
    if (valid) {
        if (ph != 0u) {
            const uint2 st = ckc[(size_t)(ph - 1u) * im.nsub_cap + i];
            p = st.x;
            z = st.y & 63u;
            m = (st.y >> 6) & 7u;
        } else {
            uint32_t zm = 0;
            if (i > 0) {
                const uint4 pe = P.substate[g - 1];
                p = pe.x;
                zm = pe.y;
            }
            const uint4 me = P.substate[g];
            if (me.z != p || me.w != zm || p < i * S || p >= i * S + 32u) {
                flags |= HJD_FLAG_NOT_CONVERGED; // host re-synchronises
                go = false;
            }
            z = zm & 0xFFu;
            m = zm >> 8;
        }
    }

    WriteCtx W = {s_ring + tid, P.rec + im.rec_off, im.rec_cap, P.blk + im.blk_off, im.nblocks, s_slot, 0xFFFFFFFFu, 0u, 0u, s_ent, base_b, base_r};
    W.tag = (blk & 127u) << 22; // records before the first DC symbol belong to a block begun in an earlier span
    const uint32_t nrec0 = nrec;
    const uint32_t row_bit0 = i * S;
    const uint32_t pend = min(row_bit0 + (ph + WWN) * P.phase_words * 32u, min(row_bit0 + S, total_bits));
    if (go && p < pend && blk < im.nblocks) {
        const EntropyLds L = {s_lut, s_sec, s_mt, lds_addr(s_mt4), lds_addr(s_sec), lds_addr(s_slot)};
        const uint32_t tabY = table_pair(ts->slot_pack, 0, 2u), tabCb = table_pair(ts->slot_pack, 1, 2u), tabCr = table_pair(ts->slot_pack, 2, 2u);
        const uint32_t *rowp = s_rows + (i - wg * G);
        RestartCtx rc = {P.mpos + im.mark_off, 0u, 0u, im.ri_blocks, im.ri_blocks ? im.ri_blocks - blk % im.ri_blocks : 0u, im.conformant};
        if (im.restart) {
            rc.nmark = min(P.st[ii].nmarkers, im.mark_cap);
            rc.mk = min(P.submk[g], rc.nmark); // first marker past the subsequence's first bit (k_sync); write_span moves it past p
            write_span<true, WGSZ, ENTCAP>(L, rowp, stride, row_bit0, tabY, tabCb, tabCr, im.nby, im.nb, p, z, m, pend, blk, nrec, W, flags, rc);
        } else {
            write_window_plain<WGSZ, ENTCAP>(lds_addr(s_mt4), lds_addr(s_sec), lds_addr(s_slot), rowp, stride, row_bit0, p, z, m, pend, blk, nrec, W, flags);
        }
    }
    // the span must end in the state, and with the record count, its checkpoints hold (else the places of everything
    // after it are wrong: the host re-synchronises) — unless the image was complete before the span's end
    if (go && blk < im.nblocks && !(flags & (HJD_FLAG_BAD_CODE | HJD_FLAG_OVERRUN)) &&
        (p != mine.x || (z | (m << 6)) != (mine.y & 511u) || nrec - nrec0 != my_recs))
        flags |= HJD_FLAG_NOT_CONVERGED;
    if (go && blk < im.nblocks && W.bfirst != 0xFFFFFFFFu) write_block_entry<ENTCAP>(W, blk); // the span ends inside a block it owns
    if (go && (nrec & (HJD_REC_RUN - 1u))) { // last, partial burst: records first, padding after
        const uint32_t k = nrec & (HJD_REC_RUN - 1u);
        for (uint32_t q = k; q < HJD_REC_RUN; q++) W.ring[q * WGSZ] = HJD_REC_PAD;
        if (nrec + HJD_REC_RUN - k <= im.rec_cap) {
            uint32_t *dst = W.rec + (nrec - k);
#pragma unroll
            for (uint32_t q = 0; q < HJD_REC_RUN / 4u; q++)
                store_records(dst + 4u * q, W.ring[(4u * q) * WGSZ], W.ring[(4u * q + 1u) * WGSZ], W.ring[(4u * q + 2u) * WGSZ],
                              W.ring[(4u * q + 3u) * WGSZ]);
        } else {
            flags |= HJD_FLAG_OVERRUN;
        }
    }
    // one atomic per wave
    for (int d = 32; d >= 1; d >>= 1) flags |= (uint32_t)__shfl_xor((int)flags, d, 64);
    if (lane == 0 && flags) atomicOr(&P.st[ii].flags, flags);
    // the workgroup's block entries: blocks [base_b, base_b + tot_b] (the last one in progress at the workgroup's end), those whose DC
    // symbol lies in this workgroup; 8 bytes each, consecutive lanes consecutive blocks
    __syncthreads();
    {
        uint2 *blk_out = P.blk + im.blk_off + base_b;
        const uint32_t n = min(tot_b + 1u, ENTCAP);
        for (uint32_t k = tid; k < n; k += WGSZ) {
            const uint32_t v = s_ent[k];
            if (v != 0xFFFFFFFFu) blk_out[k] = make_uint2(base_r + (v & 0xFFFFu), v >> 16);
        }
    }
}

__global__ __launch_bounds__(HJD_WRITE_WG) void k_write(HjdParams P) { write_body<HJD_WRITE_WG, HJD_ENT_CAP>(P); }
__global__ __launch_bounds__(HJD_WRITE_WG_SMALL) void k_write_small(HjdParams P) { write_body<HJD_WRITE_WG_SMALL, HJD_ENT_CAP_SMALL>(P); }

// ---- launch wrappers --------------------------------------------------------------------------

// Dynamic LDS above 32 KB needs the function attribute raised once per (kernel, device) — not on every launch
// (round 2 paid a hipFuncSetAttribute per launch, 36 per bench step).  `which` = 0 k_sync, 1 k_repair, 2 k_repair_totals, 3 k_write, 4 k_sync_small.
static int set_dyn_lds(int which, const void *fn, uint32_t bytes)
{
    // hipFuncSetAttribute is last-writer-wins, so the check, the call and the record of what was granted are ONE critical
    // section: two host threads launching the same kernel for the first time with different sizes (decoders with different
    // subsequence sizes or staging windows) must not leave the attribute below what the table says was granted.
    static std::mutex mu;
    static uint32_t granted[8][16];
    if (bytes <= 32768) return 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return -1;
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 16 && granted[which][dev] >= bytes) return 0;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return -1;
    if (dev < 16) granted[which][dev] = bytes;
    return 0;
}

int hjd_launch_sync(const HjdParams &P, uint32_t nwg, int mode, hipStream_t s)
{
    if (!nwg) return 0;
    if (mode == 0) {
        if (set_dyn_lds(0, reinterpret_cast<const void *>(k_sync), P.lds_bytes_sync)) return -1;
        if (P.tk_n) {
            if (set_dyn_lds(5, reinterpret_cast<const void *>(k_sync_sib), P.lds_bytes_sync + HJD_SIB_LDS_EXTRA)) return -1;
            HjdParams Q = P;
            Q.tk_nwg = nwg; // the grid: tk_n sibling workgroups per synchronisation workgroup first, then the workgroups proper
            hipLaunchKernelGGL(k_sync_sib, dim3(nwg * (P.tk_n + 1u)), dim3(HJD_SYNC_THREADS), P.lds_bytes_sync + HJD_SIB_LDS_EXTRA, s, Q);
        } else {
            hipLaunchKernelGGL(k_sync, dim3(nwg), dim3(HJD_SYNC_THREADS), P.lds_bytes_sync, s, P);
        }
    } else if (mode == 3) {
        if (set_dyn_lds(4, reinterpret_cast<const void *>(k_sync_small), P.lds_bytes_sync_small)) return -1;
        hipLaunchKernelGGL(k_sync_small, dim3(nwg), dim3(HJD_SYNC_WG_SMALL / HJD_SYNC_C), P.lds_bytes_sync_small, s, P);
    } else if (mode == 1) {
        if (set_dyn_lds(1, reinterpret_cast<const void *>(k_repair), P.lds_bytes_sync)) return -1;
        hipLaunchKernelGGL(k_repair, dim3(nwg), dim3(HJD_SYNC_THREADS), P.lds_bytes_sync, s, P);
    } else {
        if (set_dyn_lds(2, reinterpret_cast<const void *>(k_repair_totals), P.lds_bytes_sync)) return -1;
        hipLaunchKernelGGL(k_repair_totals, dim3(nwg), dim3(HJD_SYNC_THREADS), P.lds_bytes_sync, s, P);
    }
    hipError_t e_ = hipGetLastError();
    return e_ == hipSuccess ? 0 : (int)e_;
}

int hjd_launch_sync_seq(const HjdParams &P, uint32_t nwg, hipStream_t s)
{
    if (!nwg) return 0;
    if (set_dyn_lds(7, reinterpret_cast<const void *>(k_sync_seq), P.lds_bytes_sync_seq)) return -1;
    hipLaunchKernelGGL(k_sync_seq, dim3(nwg), dim3(64), P.lds_bytes_sync_seq, s, P);
    hipError_t e_ = hipGetLastError();
    return e_ == hipSuccess ? 0 : (int)e_;
}

int hjd_launch_write(const HjdParams &P, uint32_t nwg, hipStream_t s, int small)
{
    if (!nwg) return 0;
    if (small) {
        if (set_dyn_lds(6, reinterpret_cast<const void *>(k_write_small), P.lds_bytes_write_small)) return -1;
        hipLaunchKernelGGL(k_write_small, dim3(nwg), dim3(HJD_WRITE_WG_SMALL), P.lds_bytes_write_small, s, P);
        hipError_t e2_ = hipGetLastError();
        return e2_ == hipSuccess ? 0 : (int)e2_;
    }
    if (set_dyn_lds(3, reinterpret_cast<const void *>(k_write), P.lds_bytes_write)) return -1;
    hipLaunchKernelGGL(k_write, dim3(nwg), dim3(HJD_WRITE_WG), P.lds_bytes_write, s, P);
    hipError_t e_ = hipGetLastError();
    return e_ == hipSuccess ? 0 : (int)e_;
}
