// hjd_kernels.h — kernel argument bundle and launch entry points (host <-> csrc/*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hjd_device.h"

struct HjdParams {
    const HjdImage *img;       // [nimg]
    HjdImageState *st;         // [nimg]
    const HjdTableSet *tabs;   // distinct Huffman configurations
    const uint8_t *raw;        // stuffed scans
    uint8_t *ustream;          // unstuffed scans, MSB-first words
    uint32_t *chunk_kept;      // kept bytes per unstuff chunk
    uint32_t *chunk_mark;      // markers per unstuff chunk (restart mode)
    uint32_t *mpos;            // marker bit positions
    uint8_t *mval;             // marker codes
    uint4 *substate;           // per subsequence: end state, start state used
    uint32_t *subcnt;          // per subsequence: blocks | records << 16
    uint32_t *submk;           // per subsequence of a restart stream: index of the first marker past its first bit
    unsigned long long *dbg;   // diagnostics (HJD_SYNC_PROF=1): 32 words per synchronisation workgroup, else null
    uint2 *ckpt;               // checkpoints, per image [window][subsequence]: x = bit position at the window's end, y = z | m << 6 |
                               // blocks inside the window << 9 | records inside it << 20; after ck_scratch entries: k_sync's
                               // private copies for its warm-up slots, per workgroup [window][slot < sync_overlap]
    uint64_t ck_scratch;       // first entry of that scratch area
    // Sibling trajectories (small batches: the chip has room for several decodes of every subsequence).  tk_n sibling workgroups
    // per synchronisation workgroup repeat round 0 from the same guess with another block-in-MCU label; their
    // checkpoints (slab t at tk_ckpt + t * tk_stride, canonical layout) are further states a refinement walk may fall into step with.
    uint2 *tk_ckpt;
    uint64_t tk_stride;
    uint32_t *tk_flags;        // [synchronisation workgroup][sibling]: the run (epoch) whose checkpoints are complete
    uint32_t tk_n, tk_nwg, epoch;
    uint32_t *sync_flags;      // [large synchronisation workgroup]: the run (epoch) whose final records it has published (uniform regions across workgroup borders)
    uint32_t *rep_flags;       // [repair workgroup]: stamp (epoch << 2 | mode) of the repair launch whose records it has left (a chain per image)
    uint2 *wgsum;              // per final-pass workgroup: blocks, records (padded runs)
    uint32_t *rec;             // AC coefficient records
    uint2 *blk;                // per block: first record index, DC difference as decoded (k_write)
    int16_t *dca;              // per block: DC value after prediction
    // workgroup -> image for every kernel's workgroup numbering (global numbers, like the off_* below)
    const uint32_t *map_chunk, *map_syncwg, *map_syncwg_small, *map_repwg, *map_writewg, *map_writewg_small, *map_dcchunk;
    const uint2 *map_group;    // IDCT/colour workgroup -> (image, absolute index of its first block in blk)
    uint32_t *dcsum;           // per DC chunk: totals of the Y, Cb, Cr differences
    uint32_t ndcchunks;
    uint8_t *out;              // RGB / BMP bytes
    int nimg;
    // A launch may cover only a contiguous range of images (sub-batches on separate streams): these are
    // added to blockIdx.x so that every kernel still sees batch-global workgroup / image numbers.
    uint32_t off_chunk, off_syncwg, off_syncwg_small, off_repwg, off_writewg, off_writewg_small, off_img, off_dcchunk, off_group;
    uint32_t conformant;       // hjd_options.conformant: selects the conformant instantiation of the back end
    uint32_t gate_emulated;    // k_idct_color: 1 = only images the sequential emulation has just decoded (HjdImageState::emulated)
    const uint32_t *seq_list;  // k_sync_seq: per workgroup 64 device images (all ones = none) that share table set and sampling
    uint32_t off_seqwg;
    uint32_t lds_bytes_sync_seq;
    const uint32_t *rst_list;  // device images that carry DRI, ascending (the sequential emulation's candidates)
    const uint32_t *rst_groups; // the back-end workgroups (batch-global numbers) of those images: the grid of the gated launch
    // Segment-parallel emulation (hjd_seq.hip: k_seg_*), images of rst_list with HjdImage::seg_cap != 0:
    uint32_t nrst;             // entries of rst_list
    uint32_t seg_cut;          // blocks after which a node's segment is cut off (at least two restart intervals and 64; top bit set: exactly the low bits)
    uint32_t *seg_pos;         // candidates (raw byte offsets of FF xx, xx != 00), ascending per image
    uint32_t *seg_cnt;         // per entry of rst_list: candidates found; all ones = not decoded this way in this run
    uint32_t *seg_nchain;      // per entry of rst_list: nodes on the true chain
    uint32_t *seg_plan;        // [2][nrst + 1] exclusive prefixes of the work items: segment decode, write pass
    uint32_t *seg_next;        // per node (candidate x block-in-MCU label): next candidate + 1 | label there << 25 | how the segment ended << 29
    uint4 *seg_w;              // per node: blocks, records, DC differences (Y | Cb << 16, Cr) of its segment
    uint4 *seg_base;           // per node on the chain: the same sums over everything before it
    uint32_t *seg_chain;       // per image: the chain's nodes in order
    uint32_t subseq_bits;      // power of two, 64..4096
    uint32_t log2_words;       // log2(subseq_bits / 32)
    uint32_t sync_overlap;     // warm-up subsequences at the front of every synchronisation workgroup
    uint32_t phase_words;      // words of a subsequence staged in LDS at a time (power of two, <= subseq_bits / 32)
    uint32_t write_windows;    // staging windows one k_write lane decodes (power of two, <= windows per subsequence)
    uint32_t lds_sec_word;     // entropy kernels' dynamic LDS: word offset of the secondary tables
    uint32_t lds_state_word;   //                               word offset of the per-workgroup state arrays
    uint32_t lds_sec_word_s;   // the same two offsets for the synchronisation kernels (their primary tables are twice as large)
    uint32_t lds_state_word_s;
    uint32_t lds_bytes_sync;   //                               total bytes, k_sync / k_repair
    uint32_t lds_bytes_sync_small; //                           total bytes, k_sync_small
    uint32_t lds_bytes_write;  //                               total bytes, k_write
    uint32_t lds_bytes_write_small; //                          total bytes, k_write_small
};

int hjd_launch_unstuff(const HjdParams &P, uint32_t nchunks, hipStream_t s);
int hjd_launch_sync(const HjdParams &P, uint32_t nwg, int mode, hipStream_t s); // mode 0 k_sync, 1 k_repair, 2 k_repair_totals, 3 k_sync_small
int hjd_launch_dcscan(const HjdParams &P, uint32_t nchunks, hipStream_t s);
int hjd_launch_write(const HjdParams &P, uint32_t nwg, hipStream_t s, int small = 0);
int hjd_launch_sync_seq(const HjdParams &P, uint32_t nwg, hipStream_t s);
int hjd_launch_idct_color(const HjdParams &P, uint32_t ngroups, hipStream_t s);
int hjd_launch_seq_decode(const HjdParams &P, uint32_t first, uint32_t n, hipStream_t s, int after_seg = 0); // candidates rst_list[first .. first + n): those flagged DIRTY are decoded
int hjd_launch_seg_emulation(const HjdParams &P, hipStream_t s); // the same, parallel over the resynchronisations, for the images with node space
int hjd_launch_color_table(uint8_t *d_out, hipStream_t s); // test hook: 3 << 24 bytes
