// hjd_device.h — data layout shared by the host runtime and the gfx950 kernels.
//
// Everything a kernel reads lives in HBM slabs owned by a batch:
//   raw      : the stuffed scan bytes of all images, image i at raw_off (16-B aligned, padded)
//   ustream  : the unstuffed scans as MSB-first 32-bit words, transposed in groups of 64 subsequences
//              (word k of subsequence i at ((i/64)*(wps+2) + k)*64 + i%64, rows extended by two words),
//              image i at ustream_off
//   substate : one uint4 per entropy-decode subsequence (fixed bit length): the decoder state at its end
//              (bit position, z | m << 8) and the start state that end state was computed from
//   subcnt   : per subsequence, blocks completed inside (low 16 bits) | AC records produced inside (high 16)
//   wgsum    : per final-pass workgroup (HJD_WRITE_WG subsequences): totals of blocks and of records
//              (each subsequence's records rounded up to HJD_REC_RUN)
//   rec      : AC coefficient records in decode order (u32: value int16 | slot << 16 | (block & 127) << 22, where
//              slot = u*8+v is the coefficient's place after DeZigZag + TransformArray (loadjpg.cpp:156-180), u the
//              horizontal frequency; all ones = padding), every subsequence's run rounded up to HJD_REC_RUN
//              records so that it is written in aligned 16-byte stores
//   blk      : one uint2 per block: x = index of the block's first record, y = the decoded DC difference (int16 in the low half).
//              Written once and whole (8 bytes) by the k_write lane that decoded the block's DC symbol; read by the DC prediction
//              kernels (y) and, at the borders of its groups only, by the back end (x).  One extra entry per image closes the
//              last block's record range.  (Until round 4: 16 bytes with the mask of the slots the block's records fill, and the DC
//              difference a second time in a 2-byte array of its own — the back end now builds the masks from the records.)
//   dca      : per block, the DC value after prediction (int16), written by k_dcapply
//   wgmap    : for every kernel's workgroup numbering, workgroup -> image (one load instead of a search)
//   mpos/mval: restart mode only — bit position (in the unstuffed stream, markers removed) and second
//              byte of every marker found inside the scan
//   out      : RGB or BMP bytes
#pragma once
#include <stdint.h>

// Index widths of the primary Huffman tables, swept in round 5 (profiles/r05_lut_sweep.txt; 64 x 4K bench, same box, interleaved):
//   final pass 11 / 10 / 9 / 8 bits: 305 / 310 / 315.5 / 307 Gpixel/s (k_write alone 0.138 ms at every width).  What counts is
//   k_write's LDS: 2-byte entries, four tables + the second-level tables are 17.2 / 9.6 / 6.9 / 8.7 KB for the Annex K tables
//   (narrower primary tables need wider second-level ones: 9 is the minimum), and at 39.6 KB instead of 49.8 a final-pass
//   workgroup fits on a CU BESIDE eight back-end workgroups (8 x 14.6 KB) and four of them fit a CU instead of three.
//   synchronisation 11 / 10 / 9 bits: 305 / 301 / 301 — k_sync itself gets slower (0.286 / 0.306 / 0.322 ms: fewer two-symbol
//   entries fit a narrower window, more second look-ups) and its 81 KB are not what the other kernels wait for.
#ifndef HJD_LUT_BITS
#define HJD_LUT_BITS 9         // primary table of the final pass (k_write) and of the emulation kernels: index bits
#endif
#ifndef HJD_LUT_BITS_S
#define HJD_LUT_BITS_S 11      // ... of the synchronisation kernels (their entries are 4 bytes: 4 << HJD_LUT_BITS_S bytes of LDS per table)
#endif
#define HJD_LUT_SIZE (1 << HJD_LUT_BITS)
#define HJD_LUT_SIZE_S (1 << HJD_LUT_BITS_S)
#define HJD_SLOTS 6            // (component 0..2) * 2 + (0 = DC, 1 = AC)
#define HJD_UNSTUFF_CHUNK 4096 // raw bytes per unstuff workgroup (256 lanes x 16 B)
#ifndef HJD_SYNC_WG
#define HJD_SYNC_WG 512        // subsequences (slots, incl. warm-up) per synchronisation workgroup
#endif
#ifndef HJD_SYNC_C
#define HJD_SYNC_C 1           // subsequences every lane of a synchronisation workgroup decodes side by side (independent chains in
                               // one loop body: one chain's table look-up hides behind the other's arithmetic); the workgroup
                               // has HJD_SYNC_WG / HJD_SYNC_C lanes
#endif
#define HJD_SYNC_THREADS (HJD_SYNC_WG / HJD_SYNC_C)
#define HJD_SYNC_WG_SMALL 64   // slots of the small synchronisation workgroup: one wave, for images whose scan is at most that many
                               // subsequences less the warm-up (HjdImage::sync_small)
#ifndef HJD_WRITE_WG
#define HJD_WRITE_WG 256       // threads (= subsequences) per final-pass workgroup
#endif
#define HJD_SIB_LDS_EXTRA (HJD_SYNC_WG * 25u + 64u) // k_sync_sib: the end states of the siblings' trajectories per slot (2 bytes each), one byte of bookkeeping
#define HJD_MAX_SIBLINGS 11u    // sibling trajectories per subsequence at most (k_sync_sib: small batches): 2 x 6 labels less the canonical one
#define HJD_SEQ_MAX_BITS 65536u // scans up to this many bits go to k_sync_seq (one lane per image)
#define HJD_WRITE_WG_SMALL 64   // lanes of the small final-pass workgroup (k_write_small: images with HjdImage::sync_small)
#define HJD_ENT_CAP_SMALL 512u
#ifndef HJD_ENT_CAP
#define HJD_ENT_CAP 2048u      // block entries a final-pass workgroup collects in LDS before it writes them out together
#endif
#define HJD_GROUP_MCUS 16      // MCUs per IDCT/colour workgroup
#define HJD_DC_CHUNK 256       // MCUs per DC-prediction workgroup (one per thread)

// Huffman decode tables of one distinct (DC/AC x 3 components) configuration.
// Primary table: indexed by the next HJD_LUT_BITS bits.  Entry (u16):
//   bits 0-4 code length (1..16), bits 5-8 number of value bits s, bits 9-15 zig-zag advance:
//     DC symbol n          -> s = n, adv = 1
//     AC (r,s>0)           -> adv = r + 1 (coefficient lands at z + r)
//     AC EOB (0,0)         -> adv = 64;  ZRL (15,0) -> adv = 16;  other (r,0) -> adv = 0
//     (loadjpg.cpp:768-808)
//   code length 0: the code is longer than HJD_LUT_BITS; bits 5-15 = 1 + index of the secondary
//   table for this prefix (0 = no code has this prefix).  Secondary tables are indexed by the next
//   HJD_SEC_BITS bits and hold entries of the same format (length 0 = no such code).
// (On the q85 bench content 1.1 % of the AC symbols are longer than 10 bits and 0.2 % longer than 11: the synchronisation loop,
// whose step is a hand-scheduled chain, keeps 11 bits; the final pass, whose waves take a divergent branch or other in nearly every
// iteration anyway, loses nothing measurable to the more frequent second look-up at 9.)
#define HJD_SEC_BITS (16 - HJD_LUT_BITS)
#define HJD_SEC_SIZE (1 << HJD_SEC_BITS)
#define HJD_SEC_BITS_S (16 - HJD_LUT_BITS_S)
#define HJD_SEC_SIZE_S (1 << HJD_SEC_BITS_S)
#define HJD_MAX_SEC 512
struct HjdTableSet {
    uint32_t nlut;       // distinct primary tables in use (identical DHT payloads share one)
    uint32_t nsec;       // secondary tables in use
    uint32_t slot_pack;  // 6 x 4 bits: slot (component*2 + isAC) -> primary table index
    uint32_t nsec_s;     // secondary tables of the synchronisation form (its primary index may be narrower: other prefixes)
    uint16_t lut[HJD_SLOTS][HJD_LUT_SIZE];
    uint16_t sec[HJD_MAX_SEC][1 << HJD_SEC_BITS];
    // The same tables in the form the synchronisation pass wants (it needs no values): bits 0-4 code length + value bits
    // (0 = long code, as above), bits 5-11 zig-zag advance, bits 14-15 = AC records the step produces.
    // Two entries per index, read as ONE 32-bit word: [0] (low half) the symbol at the head of the window; [1] that symbol AND
    // the next one as a single step (lengths, advances and records added up) wherever the next code lies wholly inside the
    // HJD_LUT_BITS window and the first symbol is neither an end-of-block nor a long code — else a copy of [0].  The decoder
    // takes [1] while the first symbol cannot end the block (z < 48: an AC advance is at most 16) and the second symbol still
    // starts inside the staging window.  On the q85 bench content 38 % of the look-ups go away.
    uint16_t lut_s[HJD_SLOTS][HJD_LUT_SIZE_S][2];
    uint16_t sec_s[HJD_MAX_SEC][1 << HJD_SEC_BITS_S];
};

struct HjdImage {
    uint32_t width, height;
    uint32_t hf, vf;        // luma sampling factors
    uint32_t nby;           // hf*vf luma blocks per MCU
    uint32_t nb;            // nby + 2 blocks per MCU
    uint32_t mcux, mcuy;    // MCU grid (loadjpg.cpp:1170-1182)
    uint32_t nblocks;       // mcux*mcuy*nb
    uint32_t scan_len;      // raw scan bytes
    uint32_t tabset;        // index into the HjdTableSet array
    uint32_t chunk0;        // first unstuff chunk (global numbering)
    uint32_t nchunks;
    uint32_t sub0;          // first subsequence (global numbering, capacity-based)
    uint32_t nsub_cap;      // capacity: ceil(scan_len*8 / subseq_bits)
    uint32_t syncwg0;       // first synchronisation workgroup (overlapped tiling)
    uint32_t nsyncwg;
    uint32_t writewg0;      // first final-pass workgroup (HJD_WRITE_WG subsequences each)
    uint32_t nwritewg;
    uint32_t repwg0;        // first repair workgroup (HJD_SYNC_WG subsequences each, no overlap)
    uint32_t nrepwg;
    uint32_t dcchunk0;      // first DC-prediction chunk
    uint32_t ndcchunk;
    uint32_t group0;        // first IDCT/colour workgroup
    uint32_t groups_per_row;
    uint32_t ngroups;
    uint32_t out_stride;    // bytes per output row (3W, or BMP padded)
    uint32_t out_format;    // 0 RGB, 1 BMP
    uint32_t restart;       // DRI seen: the reference's resync checks are live (loadjpg.cpp:535-550, :631-640)
    uint32_t ri_blocks;     // blocks per restart interval (Ri * nb); 0 = not known (the reference's own host never passes Ri on,
                            // openjpg.cpp:441-446, and neither does a DRI segment that says 0): markers are taken where they are
    uint32_t mark_off;      // first entry of this image in the marker arrays
    uint32_t mark_cap;      // entries reserved
    uint32_t n_rst;         // RSTn markers a well-formed scan of this size holds: ceil(nmcu/Ri) - 1 (Ri not known: the pairs the
                            // host counted in the scan before EOI)
    uint32_t rec_cap;       // records reserved
    uint32_t conformant;    // hjd_options.conformant: DC reset at RSTn, true-cosine rounding IDCT, rounding colour
    uint32_t sync_small;    // 1: the image's one synchronisation workgroup is a small one (k_sync_small, its own numbering);
                            // 2: none at all — a lane of k_sync_seq decodes the whole scan (thumbnails)
    uint32_t write_small;   // 1: its final-pass workgroups are small ones (k_write_small, their own numbering)
    uint32_t wgsum0;        // first entry of the image in wgsum[] (one array for both kinds of final-pass workgroups)
    uint32_t seg_cap;       // segment-parallel emulation (hjd_seq.hip): candidates reserved (0: none — the image is left to k_seq_decode),
    uint32_t seg_off;       //   first entry in seg_pos,
    uint32_t seg_node;      //   first node ((seg_cap + 1) * nb nodes) in seg_next / seg_w / seg_base / seg_chain
    uint32_t seg_ri;        //   blocks per restart interval for the cut-off length of a segment: ri_blocks, or (Ri not known) the
                            //   image's blocks over the marker pairs the host counted
    uint64_t raw_off;       // byte offset into raw slab
    uint64_t ustream_off;   // byte offset into ustream slab (multiple of 16)
    uint64_t blk_off;       // first entry of this image in blk (nblocks + 1 entries per image)
    uint64_t rec_off;       // first record
    uint64_t out_off;       // byte offset into out slab
    uint8_t qt[3][64];      // dequantisation tables as USED per component (Cb already holds Cr's), indexed by SLOT u*8+v
};

// zig-zag index -> slot u*8+v (u = horizontal frequency): DeZigZag (natural index v*8+u, table loadjpg.cpp:56-66)
// followed by TransformArray (loadjpg.cpp:167-180)
#define HJD_ZZ_TO_SLOT                                                                                                   \
    {0,  8,  1,  2,  9,  16, 24, 17, 10, 3,  4,  11, 18, 25, 32, 40, 33, 26, 19, 12, 5,  6,  13, 20, 27, 34, 41, 48, 56, 49, 42, 35, \
     28, 21, 14, 7,  15, 22, 29, 36, 43, 50, 57, 58, 51, 44, 37, 30, 23, 31, 38, 45, 52, 59, 60, 53, 46, 39, 47, 54, 61, 62, 55, 63}

// per image, written by the kernels
struct HjdImageState {
    uint32_t ustream_len;   // unstuffed bytes
    uint32_t flags;         // HJD_FLAG_*
    uint32_t total_blocks;  // blocks completed over all subsequences
    uint32_t sync_iters;    // diagnostics: sum over workgroups and launches of refinement rounds
    uint32_t sync_decodes;  // diagnostics: subsequence decodes executed by the synchronisation launches
    uint32_t sync_iters_max;
    uint32_t nmarkers;      // restart mode: markers (FF xx, xx != 0) found inside the scan
    uint32_t emulated;      // the emulation (hjd_seq.hip) has decoded this image in this run: 1 = one lane, 2 = parallel over the resynchronisations
};
#define HJD_FLAG_BAD_CODE 1u      // no Huffman code matched on the true path
#define HJD_FLAG_OVERRUN 2u       // zig-zag index ran past 63 (loadjpg.cpp:780-806)
#define HJD_FLAG_NOT_CONVERGED 4u // a subsequence was decoded from a stale start state
#define HJD_FLAG_SHORT 8u         // stream ended before all blocks were decoded
#define HJD_FLAG_DIRTY 16u        // restart stream outside the clean-resync domain, found while decoding
#define HJD_FLAG_BAD_MARKER 32u   // the same verdict from the unstuff stage (a non-RSTn marker between intervals): unlike the
                                  // decode-stage flags it is not recomputed by a host re-synchronisation and must survive it

// synchronisation record of one subsequence (substate[]): x = end state bit position (>= end of the
// subsequence), y = end state z | m << 8, z / w = the start state this record was computed from
#define HJD_REC_PAD 0xFFFFFFFFu
#ifndef HJD_REC_RUN
#define HJD_REC_RUN 8
#endif
// every k_write lane's run of records is padded to a multiple: a lane stores 32 aligned bytes at a time — a whole
                      // sector (16-byte bursts, round 2, wrote every sector of the record array twice)
