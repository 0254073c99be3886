// hjd_entropy.hip — Huffman VLC decode of the scan on gfx950 (MI355X, CDNA4).
//   k_sync    ProcessHuffmanBlock, speculative: find the decoder state at every subsequence start
//   k_repair  the same machinery re-run only where a wave's warm-up was too short
//   k_write   ProcessHuffmanBlock, final pass: coefficients out            (loadjpg.cpp:499-863)
//
// The scan of a baseline JPEG is one bit-serial string.  It is cut into fixed-size SUBSEQUENCES
// (S bits), one per lane.  Huffman streams self-synchronise: a decoder started at a wrong
// position/state soon falls into step with the true decode.  Each lane decodes its subsequence
// from a guessed start state, then again from the end state of its predecessor, until
// start == predecessor's end everywhere in the workgroup.  Workgroups overlap by a few warm-up
// subsequences, so the first owned lane of every workgroup has (with overwhelming probability)
// the true state; k_write re-checks start == predecessor's end for EVERY subsequence across
// workgroup borders, and the host re-synchronises if that induction step ever fails.  Outputs
// are written only by k_write, so speculation never reaches them.
//
// Stream layout (written by k_unstuff_write): TRANSPOSED in groups of 64 subsequences — word k of
// subsequence i lives at word ((i/64)*(wps+2) + k)*64 + i%64, and every row is extended by the
// first two words of the next row — so that a wave stages its 64 rows with fully coalesced loads and
// then reads them from LDS without bank conflicts ([k*64 + lane]).
// LDS per workgroup: the Huffman tables (primary 10-bit tables + secondary tables for longer codes,
// shared by its 4 waves) and each wave's rows.  Waves are independent after the table load: each wave
// owns 64 - K subsequences plus K warm-up subsequences before them and iterates with __shfl_up /
// __ballot.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hjd_device.h"
#include "hjd_kernels.h"
#include "hjd_kutil.h"

struct EntropyLds {
    const uint16_t *lut; // [nlut][1024]
    const uint16_t *sec; // [nsec][HJD_SEC_SIZE]
};

// address of word 0 of subsequence (row) i in the transposed stream; successive words are 64 words apart
__device__ __forceinline__ const uint32_t *row_base(const uint32_t *w, uint32_t i, uint32_t lw)
{
    return w + ((size_t)(i >> 6) * ((1u << lw) + 2u)) * 64u + (i & 63u);
}

__device__ __forceinline__ void load_tables(const HjdTableSet *ts, uint16_t *s_lut, uint16_t *s_sec)
{
    const uint32_t nl = ts->nlut * (HJD_LUT_SIZE * 2 / 16), ns = ts->nsec * (HJD_SEC_SIZE * 2 / 16);
    const uint4 *src = reinterpret_cast<const uint4 *>(&ts->lut[0][0]);
    uint4 *dst = reinterpret_cast<uint4 *>(s_lut);
    for (uint32_t k = threadIdx.x; k < nl; k += blockDim.x) dst[k] = src[k];
    src = reinterpret_cast<const uint4 *>(&ts->sec[0][0]);
    dst = reinterpret_cast<uint4 *>(s_sec);
    for (uint32_t k = threadIdx.x; k < ns; k += blockDim.x) dst[k] = src[k];
}

// (DC | AC << 16) byte offsets of a component's primary tables, from the packed slot map
__device__ __forceinline__ uint32_t table_pair(uint32_t slot_pack, uint32_t comp)
{
    const uint32_t dc = ((slot_pack >> (comp * 8u)) & 7u) * (HJD_LUT_SIZE * 2u);
    const uint32_t ac = ((slot_pack >> (comp * 8u + 4u)) & 7u) * (HJD_LUT_SIZE * 2u);
    return dc | (ac << 16);
}

// ---- bit readers ------------------------------------------------------------------------------
// window() = the next 32 bits; advance(n) consumes n <= 31 bits.  State: words hi, lo and sh = number
// of unconsumed bits of hi (0..31); window = alignbit(hi, lo, sh).

// Both kernels stage the wave's 64 rows in LDS first ([k*64 + lane], coalesced loads), so that the
// loop's only vector-memory operations are k_write's record stores, which it never waits for (loads
// and stores share one counter on gfx9-class hardware; reading the stream from L1 inside the loop
// was measured 2x slower for k_write and no faster for k_sync).
struct ReaderLds {
    uint32_t hi, lo, nxt, sh;
    const uint32_t *nx;
    __device__ __forceinline__ void init(const uint32_t *rowp, uint32_t rel)
    {
        const uint32_t k0 = rel >> 5, o0 = rel & 31u;
        const uint32_t *sp = rowp + k0 * 64u;
        const uint32_t a = sp[0], b = sp[64];
        hi = a;
        lo = o0 ? b : a;
        nx = sp + (o0 ? 128 : 64);
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
        nx += roll ? 64 : 0;
        sh = (uint32_t)sh2 & 31u;
    }
};

// Decode symbols from state (p,z,m) until p >= end.  One symbol per iteration.
//   WRITE = false: nblk counts blocks completed, nrec the AC records they hold      (synchronisation)
//   WRITE = true : nblk is the absolute block index, nrec the index of the next record.  Every non-zero
//                  AC coefficient becomes one record (value | zig-zag slot << 16: the DCT[] array of
//                  loadjpg.cpp:499-863 in sparse form), appended to the lane's own run and stored four
//                  at a time; boff[b] = first record of block b; DC DIFFERENCES go to dcd[] — DC
//                  prediction (loadjpg.cpp:619-667) is a prefix sum done by k_dcsum/k_dcapply
// `rowp` points at word 0 of the lane's row (64 words between successive words), `row_bit0` is the
// absolute bit position of the row's first bit.  tabY/tabCb/tabCr: byte offsets of the component's
// (DC | AC << 16) primary tables.
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
    uint32_t mk;          // index of the next marker (first position > p at the start)
    uint32_t ri_blocks;   // blocks per interval, 0 = no intervals expected
    uint32_t left;        // WRITE: blocks left in the current interval
};

template <bool WRITE, bool RESTART, class Reader>
__device__ __forceinline__ void decode_span(const EntropyLds &L, const uint32_t *__restrict__ rowp, uint32_t row_bit0,
                                            uint32_t tabY, uint32_t tabCb, uint32_t tabCr, uint32_t nby, uint32_t nb,
                                            uint32_t &p, uint32_t &z, uint32_t &m, uint32_t end, uint32_t &nblk,
                                            uint32_t &nrec, uint32_t *__restrict__ rec, uint32_t rec_cap,
                                            uint32_t *__restrict__ boff, int16_t *__restrict__ dcd, uint32_t nblocks,
                                            uint32_t &flags, RestartCtx rc, uint32_t *s_rbuf = nullptr)
{
    // WRITE: s_rbuf[k * HJD_SYNC_WG] (k < HJD_REC_RUN) = this lane's records waiting for their store burst.
    // A full burst leaves as back-to-back 16-byte stores to one aligned 32-byte piece of the record
    // array, so that the memory side sees whole sectors instead of a trickle of partial ones.
    Reader rd;
    rd.init(rowp, p - row_bit0);
    uint32_t mnext = 0xFFFFFFFFu;
    if (RESTART) {
        while (rc.mk < rc.nmark && rc.mpos[rc.mk] <= p) rc.mk++;
        mnext = rc.mk < rc.nmark ? rc.mpos[rc.mk] : 0xFFFFFFFFu;
    }
    uint32_t pair = m < nby ? tabY : (m == nby ? tabCb : tabCr);
    uint32_t tb = z ? pair >> 16 : pair & 0xFFFFu;
    const char *lutb = reinterpret_cast<const char *>(L.lut);

    while (p < end && (!WRITE || nblk < nblocks)) {
        rd.prefetch();
        const uint32_t win = rd.window();
        uint32_t e = *reinterpret_cast<const uint16_t *>(lutb + tb + ((win >> (32 - HJD_LUT_BITS)) << 1));
        if ((e & 31u) == 0u) {
            // code longer than HJD_LUT_BITS: second-level table on the next HJD_SEC_BITS bits (IsInHuffmanCodes,
            // loadjpg.cpp:335-392, for the lengths the primary table cannot hold); 0 = no such code
            const uint32_t sub = e >> 5;
            e = sub ? L.sec[(sub - 1u) * HJD_SEC_SIZE + ((win >> 16) & (HJD_SEC_SIZE - 1u))] : 0u;
            if ((e & 31u) == 0u) {
                if (WRITE) flags |= HJD_FLAG_BAD_CODE;
                e = 16u; // skip 16 bits: any progress will do, the result is discarded or flagged
            }
        }
        const uint32_t len = e & 31u, s = (e >> 5) & 15u, adv = e >> 9;
        const bool isdc = z == 0u;
        if (WRITE) {
            // value bits + DetermineSign (loadjpg.cpp:396-409)
            const uint32_t vb = ((win << len) >> 1) >> (31u - s);
            const int val = (int)vb - ((vb < ((1u << s) >> 1)) ? (int)((1u << s) - 1u) : 0);
            if (isdc) {
                dcd[nblk] = (int16_t)val;
                boff[nblk] = nrec;
                // the reference looks for a marker again between the DC code and its extra bits
                if (RESTART && s != 0u && p + len + 8u > mnext) flags |= HJD_FLAG_DIRTY;
            } else if (s != 0u) { // AC run/size: loadjpg.cpp:776-808
                const uint32_t zw = z + adv - 1u;
                uint32_t w = ((uint32_t)val & 0xFFFFu) | (zw << 16);
                if (zw > 63u) {
                    flags |= HJD_FLAG_OVERRUN;
                    w = HJD_REC_PAD;
                }
                const uint32_t k = nrec & (HJD_REC_RUN - 1u);
                s_rbuf[k * HJD_SYNC_WG] = w;
                if (k == HJD_REC_RUN - 1u) {
                    if (nrec < rec_cap) {
                        uint4 *dst = reinterpret_cast<uint4 *>(rec + (nrec - (HJD_REC_RUN - 1u)));
#pragma unroll
                        for (int g = 0; g < HJD_REC_RUN / 4; g++)
                            dst[g] = make_uint4(s_rbuf[(4 * g) * HJD_SYNC_WG], s_rbuf[(4 * g + 1) * HJD_SYNC_WG],
                                                s_rbuf[(4 * g + 2) * HJD_SYNC_WG], s_rbuf[(4 * g + 3) * HJD_SYNC_WG]);
                    } else {
                        flags |= HJD_FLAG_OVERRUN;
                    }
                }
                nrec++;
            }
        } else if (!isdc && s != 0u) {
            nrec++;
        }
        z += adv;
        tb = isdc ? pair >> 16 : tb;
        const bool blk_end = z >= 64u;
        if (blk_end) { // EOB or last coefficient: next block (DecodeMCU order, loadjpg.cpp:945-997)
            z = 0u;
            nblk++;
            if (WRITE && nblk == nblocks) boff[nblk] = nrec; // end of the last block's records
            m = (m + 1u == nb) ? 0u : m + 1u;
            pair = m < nby ? tabY : (m == nby ? tabCb : tabCr);
            tb = pair & 0xFFFFu;
        }
        const uint32_t tot = len + s;
        p += tot;
        rd.advance(tot);
        if (RESTART) {
            if (WRITE && blk_end) rc.left--;
            if ((blk_end && p + 8u > mnext) || p > mnext) {
                if (WRITE && nblk < nblocks) {
                    // clean only if a whole interval of whole MCUs ends exactly here
                    const bool clean = blk_end && p <= mnext && m == 0u && (rc.ri_blocks == 0u ? false : rc.left == 0u);
                    if (!clean) flags |= HJD_FLAG_DIRTY;
                }
                p = mnext;
                z = 0u;
                m = 0u;
                pair = tabY;
                tb = pair & 0xFFFFu;
                rc.left = rc.ri_blocks;
                rc.mk++;
                mnext = rc.mk < rc.nmark ? rc.mpos[rc.mk] : 0xFFFFFFFFu;
                rd.init(rowp, p - row_bit0);
            } else if (WRITE && blk_end && rc.ri_blocks != 0u && rc.left == 0u && nblk < nblocks) {
                flags |= HJD_FLAG_DIRTY; // an interval ends here but the marker is not where the reference looks
                rc.left = rc.ri_blocks;
            }
        }
    }
    if (WRITE && (nrec & (HJD_REC_RUN - 1u))) { // last, partial burst: padded
        const uint32_t k = nrec & (HJD_REC_RUN - 1u);
        if (nrec + HJD_REC_RUN - k <= rec_cap) {
            uint4 *dst = reinterpret_cast<uint4 *>(rec + (nrec - k));
#pragma unroll
            for (int g = 0; g < HJD_REC_RUN / 4; g++) {
                uint32_t v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = (uint32_t)(4 * g + j) < k ? s_rbuf[(4 * g + j) * HJD_SYNC_WG] : HJD_REC_PAD;
                dst[g] = make_uint4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

// ---- stage 1: synchronisation ----------------------------------------------------------------
// mode 0: wave w of an image owns subsequences [w*Q, (w+1)*Q), Q = 64 - K, and additionally decodes
//         the K subsequences before them as warm-up (their records belong to wave w-1).
// mode 1: re-synchronisation (host fallback): 64 owned subsequences per wave, records reloaded,
//         lane 0 takes its start from the previous wave's last record.

template <int mode>
__device__ __forceinline__ void sync_body(const HjdParams &P)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t bid = blockIdx.x + (mode == 0 ? P.off_syncwg : P.off_writewg);
    const int ii = mode == 0 ? find_image(P.img, P.nimg, bid, [](const HjdImage &q) { return q.syncwg0; })
                             : find_image(P.img, P.nimg, bid, [](const HjdImage &q) { return q.writewg0; });
    const HjdImage &im = P.img[ii];
    const uint32_t wv = (bid - (mode == 0 ? im.syncwg0 : im.writewg0)) * (HJD_SYNC_WG / 64) + (threadIdx.x >> 6);
    const uint32_t K = mode == 0 ? P.sync_overlap : 0u, Q = 64u - K;
    const uint32_t own0 = wv * Q, own1 = own0 + Q;
    const uint32_t row0 = own0 >= K ? own0 - K : 0u;

    const uint32_t S = P.subseq_bits;
    const uint32_t total_bits = P.st[ii].ustream_len * 8u;
    const uint32_t nsub = (total_bits + S - 1) / S;
    const uint32_t i = row0 + lane;
    const bool valid = i < nsub && i < own1;
    const bool owned = valid && i >= own0;
    const uint32_t g = im.sub0 + i;

    if (mode != 0) {
        // repair pass: nothing to do unless some record's start differs from its predecessor's end
        bool need = false;
        if (valid && i > 0) {
            const uint2 pe = *reinterpret_cast<const uint2 *>(&P.sub[g - 1]);
            const HjdSub r = P.sub[g];
            need = pe.x != r.u_p || pe.y != r.u_zm;
        }
        if (!__syncthreads_or(need ? 1 : 0)) return;
    }

    uint16_t *s_lut = reinterpret_cast<uint16_t *>(s_dyn);
    uint16_t *s_sec = reinterpret_cast<uint16_t *>(s_dyn + P.lds_sec_word);
    const HjdTableSet *ts = &P.tabs[im.tabset];
    load_tables(ts, s_lut, s_sec);
    const EntropyLds L = {s_lut, s_sec};
    const uint32_t tabY = table_pair(ts->slot_pack, 0), tabCb = table_pair(ts->slot_pack, 1), tabCr = table_pair(ts->slot_pack, 2);
    const uint32_t lw = P.log2_words;

    const uint32_t sp = i * S;
    const uint32_t end = min(sp + S, total_bits);
    const uint32_t nby = im.nby, nb = im.nb;
    // stage this wave's 64 rows [row0, row0+64): they straddle at most two 64-row groups of the stream
    const uint32_t rowwords = ((1u << lw) + 2u) * 64u;
    uint32_t *s_rows = s_dyn + P.lds_stream_word + (threadIdx.x >> 6) * rowwords;
    {
        const uint32_t *wbase = reinterpret_cast<const uint32_t *>(P.ustream + im.ustream_off);
        const uint32_t *src = row_base(wbase, i, lw);
        if (valid) // a lane only ever reads its own row
            for (uint32_t k = 0; k < (1u << lw) + 2u; k++) s_rows[k * 64u + lane] = src[k * 64u];
    }
    const uint32_t *rowp = s_rows + lane;

    uint32_t up = sp, uzm = 0, ep = sp, ezm = 0, nblk = 0, nrec = 0, flags = 0;
    uint32_t ext_p = sp, ext_zm = 0; // start state offered to lane 0 from outside the wave
    RestartCtx rc0 = {P.mpos + im.mark_off, 0u, 0u, im.ri_blocks, 0u};
    if (im.restart && valid) {
        rc0.nmark = min(P.st[ii].nmarkers, im.mark_cap);
        uint32_t lo = 0, hi = rc0.nmark; // first marker past the start of this row
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (rc0.mpos[mid] <= sp) lo = mid + 1;
            else hi = mid;
        }
        rc0.mk = lo;
    }
    uint32_t ndec = 0;
    __syncthreads(); // tables are in LDS; from here on the waves run independently

    if (valid) {
        if (mode == 0) {
            uint32_t p = sp, z = 0, m = 0;
            if (im.restart) decode_span<false, true, ReaderLds>(L, rowp, sp, tabY, tabCb, tabCr, nby, nb, p, z, m, end, nblk, nrec, nullptr, 0, nullptr, nullptr, 0, flags, rc0);
            else decode_span<false, false, ReaderLds>(L, rowp, sp, tabY, tabCb, tabCr, nby, nb, p, z, m, end, nblk, nrec, nullptr, 0, nullptr, nullptr, 0, flags, rc0);
            ep = p;
            ezm = z | (m << 8);
            ndec = 1;
        } else {
            const HjdSub r = P.sub[g];
            ep = r.e_p;
            ezm = r.e_zm;
            up = r.u_p;
            uzm = r.u_zm;
            nblk = r.nblk;
            nrec = r.nrec;
            if (lane == 0 && i > 0) {
                const uint2 pe = *reinterpret_cast<const uint2 *>(&P.sub[g - 1]);
                if (pe.x >= sp && pe.x < sp + 32u) { // a record of this run always ends inside the next word
                    ext_p = pe.x;
                    ext_zm = pe.y;
                }
            }
        }
    }

    uint32_t iters = 0;
    for (; iters <= 64u; iters++) {
        uint32_t wp = __shfl_up(ep, 1, 64), wzm = __shfl_up(ezm, 1, 64);
        if (lane == 0) {
            wp = ext_p;
            wzm = ext_zm;
        }
        if (i == 0) {
            wp = 0;
            wzm = 0; // the true start of the scan (loadjpg.cpp:1148-1149)
        }
        const bool changed = valid && (wp != up || wzm != uzm);
        if (!__ballot(changed)) break;
        if (iters == 1) __builtin_amdgcn_s_setprio(2); // chain tails are the critical path: let them issue first
        if (changed) {
            uint32_t p = wp, z = wzm & 0xFF, m = wzm >> 8;
            nblk = 0;
            nrec = 0;
            if (im.restart) decode_span<false, true, ReaderLds>(L, rowp, sp, tabY, tabCb, tabCr, nby, nb, p, z, m, end, nblk, nrec, nullptr, 0, nullptr, nullptr, 0, flags, rc0);
            else decode_span<false, false, ReaderLds>(L, rowp, sp, tabY, tabCb, tabCr, nby, nb, p, z, m, end, nblk, nrec, nullptr, 0, nullptr, nullptr, 0, flags, rc0);
            ep = p;
            ezm = z | (m << 8);
            up = wp;
            uzm = wzm;
            ndec++;
        }
    }

    if (owned) {
        HjdSub r;
        r.e_p = ep;
        r.e_zm = ezm;
        r.u_p = up;
        r.u_zm = uzm;
        r.nblk = nblk;
        r.nrec = nrec;
        r.pad[0] = r.pad[1] = 0;
        P.sub[g] = r;
    }
    // diagnostics
    atomicAdd(&P.st[ii].sync_decodes, ndec);
    if (lane == 0) {
        atomicAdd(&P.st[ii].sync_iters, iters);
        atomicMax(&P.st[ii].sync_iters_max, iters);
    }
}

__global__ __launch_bounds__(HJD_SYNC_WG) void k_sync(HjdParams P) { sync_body<0>(P); }

// the repair pass is its own kernel symbol so that profiles tell the two apart
__global__ __launch_bounds__(HJD_SYNC_WG) void k_repair(HjdParams P) { sync_body<1>(P); }

// ---- stage 3: final entropy decode, coefficients out ------------------------------------------

__global__ __launch_bounds__(HJD_SYNC_WG) void k_write(HjdParams P)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];

    const uint32_t bid = blockIdx.x + P.off_writewg;
    const int ii = find_image(P.img, P.nimg, bid, [](const HjdImage &q) { return q.writewg0; });
    const HjdImage &im = P.img[ii];

    uint16_t *s_lut = reinterpret_cast<uint16_t *>(s_dyn);
    uint16_t *s_sec = reinterpret_cast<uint16_t *>(s_dyn + P.lds_sec_word);
    const HjdTableSet *ts = &P.tabs[im.tabset];
    load_tables(ts, s_lut, s_sec);
    const EntropyLds L = {s_lut, s_sec};
    const uint32_t tabY = table_pair(ts->slot_pack, 0), tabCb = table_pair(ts->slot_pack, 1), tabCr = table_pair(ts->slot_pack, 2);

    const uint32_t S = P.subseq_bits;
    const uint32_t total_bits = P.st[ii].ustream_len * 8u;
    const uint32_t nsub = (total_bits + S - 1) / S;
    const uint32_t i = (bid - im.writewg0) * HJD_SYNC_WG + threadIdx.x;
    // stage this wave's group of 64 rows: (wps+2) x 256 contiguous bytes of the transposed stream
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t rowwords = ((1u << P.log2_words) + 2u) * 64u;
    uint32_t *s_rows = s_dyn + P.lds_stream_word + wv * rowwords;
    uint32_t *s_rbuf = s_dyn + P.lds_stream_word + (HJD_SYNC_WG / 64) * rowwords + threadIdx.x;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(
            row_base(reinterpret_cast<const uint32_t *>(P.ustream + im.ustream_off), i & ~63u, P.log2_words));
        uint4 *dst = reinterpret_cast<uint4 *>(s_rows);
        for (uint32_t q = lane; q < rowwords / 4; q += 64) dst[q] = src[q];
    }
    __syncthreads();
    if (i >= nsub) return;
    const uint32_t g = im.sub0 + i;
    const uint32_t end = min(i * S + S, total_bits);
    const uint32_t *rowp = s_rows + lane;

    uint32_t p = 0, zm = 0;
    if (i > 0) {
        const uint2 pe = *reinterpret_cast<const uint2 *>(&P.sub[g - 1]);
        p = pe.x;
        zm = pe.y;
    }
    uint32_t flags = 0;
    const HjdSub me = P.sub[g];
    if (me.u_p != p || me.u_zm != zm || p < i * S || p >= i * S + 32u)
        flags |= HJD_FLAG_NOT_CONVERGED; // induction step failed: host re-synchronises
    const HjdSubStart st = P.substart[g];
    uint32_t z = zm & 0xFF, m = zm >> 8, blk = st.blk0, nrec = st.rec0;
    if (!(flags & HJD_FLAG_NOT_CONVERGED)) {
        RestartCtx rc0 = {P.mpos + im.mark_off, 0u, 0u, im.ri_blocks, 0u};
        if (im.restart) {
            rc0.nmark = min(P.st[ii].nmarkers, im.mark_cap);
            uint32_t lo = 0, hi = rc0.nmark;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (rc0.mpos[mid] <= i * S) lo = mid + 1;
                else hi = mid;
            }
            rc0.mk = lo;
            rc0.left = im.ri_blocks ? im.ri_blocks - blk % im.ri_blocks : 0u;
            decode_span<true, true, ReaderLds>(L, rowp, i * S, tabY, tabCb, tabCr, im.nby, im.nb, p, z, m, end, blk,
                                               nrec, P.rec + im.rec_off, im.rec_cap, P.boff + im.boff_off,
                                               P.dcd + im.blk_off, im.nblocks, flags, rc0, s_rbuf);
        } else {
            decode_span<true, false, ReaderLds>(L, rowp, i * S, tabY, tabCb, tabCr, im.nby, im.nb, p, z, m, end, blk,
                                                nrec, P.rec + im.rec_off, im.rec_cap, P.boff + im.boff_off,
                                                P.dcd + im.blk_off, im.nblocks, flags, rc0, s_rbuf);
        }
    }
    if (flags) atomicOr(&P.st[ii].flags, flags);
}

// ---- launch wrappers --------------------------------------------------------------------------

static int set_dyn_lds(const void *fn, uint32_t bytes)
{
    if (bytes > 32768 &&
        hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        return -1;
    return 0;
}

int hjd_launch_sync(const HjdParams &P, uint32_t nwg, int mode, hipStream_t s)
{
    if (!nwg) return 0;
    if (mode == 0) {
        if (set_dyn_lds(reinterpret_cast<const void *>(k_sync), P.lds_bytes)) return -1;
        hipLaunchKernelGGL(k_sync, dim3(nwg), dim3(HJD_SYNC_WG), P.lds_bytes, s, P);
    } else {
        if (set_dyn_lds(reinterpret_cast<const void *>(k_repair), P.lds_bytes)) return -1;
        hipLaunchKernelGGL(k_repair, dim3(nwg), dim3(HJD_SYNC_WG), P.lds_bytes, s, P);
    }
    hipError_t e_ = hipGetLastError();
    return e_ == hipSuccess ? 0 : (int)e_;
}

int hjd_launch_write(const HjdParams &P, uint32_t nwg, hipStream_t s)
{
    if (!nwg) return 0;
    const uint32_t lds = P.lds_bytes + HJD_REC_RUN * HJD_SYNC_WG * 4; // + the record bursts in waiting
    if (set_dyn_lds(reinterpret_cast<const void *>(k_write), lds)) return -1;
    hipLaunchKernelGGL(k_write, dim3(nwg), dim3(HJD_SYNC_WG), lds, s, P);
    hipError_t e_ = hipGetLastError();
    return e_ == hipSuccess ? 0 : (int)e_;
}
