/*
 * hjd.h — C ABI of the MI355X-native baseline-JPEG decode path.
 *
 * This is the drop-in boundary for the hot path of harutel/hls-jpeg-decoder
 * (everything reachable from JpegDecodeHW, reference src/loadjpg.cpp:1134-1190):
 * Huffman VLC decode, dequantise, 8x8 IDCT, chroma upsample, YCbCr->RGB, run as
 * hand-written HIP kernels for gfx950.  Plain pointers and sizes only.
 *
 * Every entry point names the reference interface it replaces (file:line under
 * the reference's src/).  Results are byte-identical to the reference's CPU
 * decoder inside its parity domain; outside it the calls
 * return an error status instead of emulating the reference's undefined
 * behaviour.  There is NO CPU fallback: without a usable HIP device every
 * decode call fails with HJD_ERR_NO_DEVICE.
 */
#ifndef HJD_H
#define HJD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (per call and per image) -------------------------------- */
typedef enum hjd_status {
    HJD_OK = 0,
    HJD_ERR_NOT_JPEG = 1,       /* openjpg.cpp:481-486 / :383-386: no SOI, or a marker without 0xFF */
    HJD_ERR_UNSUPPORTED = 2,    /* valid JPEG outside the reference's domain (progressive, grey, 16-bit DQT, ...) */
    HJD_ERR_CORRUPT = 3,        /* truncated segment, invalid Huffman code or coefficient overrun in the scan */
    HJD_ERR_DIRTY_RESTART = 4,  /* restart stream outside the clean-resync domain, with hjd_options.dirty_mode = 1 */
    HJD_ERR_NO_DEVICE = 5,      /* no usable HIP device / kernel image: the product path has no CPU fallback */
    HJD_ERR_ARG = 6,            /* NULL pointer, size too small, bad enum */
    HJD_ERR_IO = 7,             /* file open/read/write failed */
    HJD_ERR_INTERNAL = 8        /* HIP runtime error or synchronisation did not converge */
} hjd_status;

const char *hjd_status_string(int status);
/* status of the last failing call on this thread (for the 1/0-returning entry points) */
int hjd_last_status(void);

/* ---- (1) file in, BMP file out -------------------------------------------- *
 * Replaces: int ConvertJpgFile(char*, char*)   openjpg.h:23, openjpg.cpp:593-684.
 * Same convention: returns 1 on success, 0 on failure.  The BMP bytes follow
 * WriteBMP24 (openjpg.cpp:504-570): 54-byte header, bottom-up rows, B,G,R order,
 * rows padded to 4 bytes (pad bytes are written as zero; the reference leaves pad
 * bytes 2 and 3 undefined).                                                     */
int hjd_convert_jpg_file(const char *szJpgFileInName, const char *szBmpFileOutName);

/* ---- (2) memory in, RGB out ------------------------------------------------ *
 * Replaces: int DecodeJpgFileData(const unsigned char*, int, unsigned char**,
 *           unsigned int*, unsigned int*)   openjpg.h:25-27 / loadjpg.h:185-187
 * (declared, never defined by the reference; its body is inlined at
 * openjpg.cpp:623-655).  Callee allocates *rgbpix (top-down, R,G,B, 3*W*H
 * bytes); release it with hjd_free() — the reference's "delete[] rgbpix" cannot
 * cross a C ABI.  Returns 1 on success, 0 on failure.                          */
int hjd_decode_jpg_file_data(const unsigned char *buf, int sizeBuf, unsigned char **rgbpix,
                             unsigned int *width, unsigned int *height);
void hjd_free(void *p);

/* ---- (3) the HW/SW cut: already-parsed tables + scan bytes -> RGB ---------- *
 * Replaces: int JpegDecodeHW(stJpegData*, unsigned h, unsigned w, uchar hFactor,
 *           uchar vFactor)   loadjpg.h:180, loadjpg.cpp:1134-1190.
 * hjd_jpeg_data carries what JpegDecodeHW reads from stJpegData (loadjpg.h:157-171)
 * with pointer/length fields instead of fixed-capacity arrays.                  */
typedef struct hjd_huff_spec {
    uint8_t defined;
    uint8_t bits[17];   /* bits[1..16] = number of codes of each length (DHT payload, openjpg.cpp:261-267) */
    uint8_t vals[256];  /* symbol values in code order (openjpg.cpp:282-295) */
} hjd_huff_spec;

typedef struct hjd_jpeg_data {
    /* stComponent::m_qTable for component ids 1..3, raw DQT (zig-zag) order (openjpg.cpp:347-350).
     * Index 0 unused.  The Cb-uses-Cr's-table quirk (loadjpg.cpp:984) is applied by the decoder. */
    uint8_t qtable[4][64];
    uint8_t dc_table_index[4]; /* stComponent::dcTable_index per component id (openjpg.cpp:213) */
    uint8_t ac_table_index[4]; /* stComponent::acTable_index (openjpg.cpp:212) */
    hjd_huff_spec htdc[4];     /* stHuffmanData::m_HTDC */
    hjd_huff_spec htac[4];     /* stHuffmanData::m_HTAC */
    const uint8_t *stream;     /* stHuffmanData::m_stream: first entropy-coded byte (openjpg.cpp:220-223) */
    size_t stream_len;
    int restart_interval;      /* stHuffmanData::m_restart_interval: only tested ">0" (loadjpg.cpp:535,631) */
    unsigned restart_mcus;     /* DRI payload Ri when the caller knows it; 0 = not known — the reference never stores it
                                * (openjpg.cpp:441-446), so its unmodified host passes 0: the markers are then taken where the scan
                                * holds them, exactly as the reference's own checks do (it never counts MCUs, loadjpg.cpp:535-550) */
    /* frame header (stImageInfo, openjpg.h:11-17) */
    unsigned width, height;
    uint8_t hfactor, vfactor;  /* m_hFactor[cY], m_vFactor[cY] */
    size_t header_bytes;
} hjd_jpeg_data;

/* Host header walk.  Replaces JpegParseHeader / ParseJFIF (openjpg.cpp:371-496).
 * Fills `out`; out->stream points INTO buf. */
int hjd_parse_header(const uint8_t *buf, size_t size, hjd_jpeg_data *out);
/* The same walk with flags: HJD_PARSE_CONFORMANT admits what only the reference cannot take (DC Huffman codes of
 * 16 bits, loadjpg.cpp:562) — used by decoders created with hjd_options.conformant. */
#define HJD_PARSE_CONFORMANT 1
int hjd_parse_header_ex(const uint8_t *buf, size_t size, hjd_jpeg_data *out, int flags);

/* Same argument order as the reference: (jdata, height, width, hFactor, vFactor).
 * rgb_out: host buffer of 3*width*height bytes (stJpegData::m_rgb).  Returns an hjd_status
 * (the reference returns 0 always). */
int hjd_jpeg_decode_hw(const hjd_jpeg_data *jdata, unsigned int jpeg_img_height, unsigned int jpeg_img_width,
                       unsigned char hFactor, unsigned char vFactor, unsigned char *rgb_out);

/* ---- (4) batch API (new; the reference decodes one image per process) ------ */
typedef struct hjd_decoder hjd_decoder; /* one per (process, device): kernels, LUT cache, workspace */
typedef struct hjd_batch hjd_batch;     /* N images resident in HBM, ready to decode */

typedef enum hjd_out_format {
    HJD_OUT_RGB = 0, /* stJpegData::m_rgb layout: top-down, R,G,B, stride 3*W */
    HJD_OUT_BMP = 1  /* WriteBMP24 file image (openjpg.cpp:504-570) */
} hjd_out_format;

typedef struct hjd_options {
    int subseq_bits;     /* entropy-decode subsequence size in bits (power of two, 64..16384); 0 = chosen per batch from its scan bytes */
    int sync_launches;   /* 1 + in-stream repair launches after the synchronisation launch; 0 = default (1: the final pass checks
                          * every start state and hjd_batch_finish repairs from the host if one is ever stale) */
    int streams;         /* image ranges a batch is split into.  The ranges are dealt round robin to at most four HIP streams (the caller's
                          * and three of the decoder's, forked from and joined to the caller's inside every run) and run one behind the
                          * other on each: one range's latency-bound synchronisation overlaps another's IDCT.  With three or more caller
                          * streams in flight on the decoder a batch runs whole, on the caller's stream.  0 = by batch size (1 below
                          * 60 Mpixel, 2 up to 400, above that ranges of about 130 Mpixel, at most 16), 1 = no split */
    int sync_overlap;    /* warm-up subsequences every synchronisation wave decodes before the ones it owns:
                          * 0 = default (16384 bits' worth, at least 4, at most 32), -1 = none, 1..48 = that many.  Too little warm-up is
                          * never wrong, only slower: hjd_batch_finish then re-synchronises from the host. */
    int dirty_mode;      /* restart streams on which the reference leaves the clean-resync domain:
                          * 0 = decoded the way the reference decodes them, by the emulation kernels (csrc/hjd_seq.hip: exact; parallel over
                          *     the places where the reference resynchronises, one lane per image as the fallback);
                          * 1 = refused with HJD_ERR_DIRTY_RESTART */
    int conformant;      /* SURVEY §8f N4, outside the parity contract, never the default: 1 = decode as a conforming baseline
                          * decoder would where the reference deviates — DC predictors reset at every RSTn, restart intervals
                          * taken by the standard's rule (no DIRTY verdicts), DC codes of 16 bits accepted, IDCT with the
                          * true cosines and round-to-nearest, colour conversion with the JFIF constants and rounding.
                          * Chroma stays nearest-neighbour.  0 = byte-identical to the reference. */
    int debug_flags;     /* test switches (HJD_DBG_*): they select among paths that must all give the same bytes; 0 in production */
    int reserved;
} hjd_options;
/* hjd_options.debug_flags */
#define HJD_DBG_NO_SMALL_SYNC 1     /* no one-wave synchronisation / final-pass workgroups for small images */
#define HJD_DBG_NO_SEQ_SYNC 2       /* no one-lane-per-image synchronisation of thumbnails */
#define HJD_DBG_NO_SEG_EMULATION 4  /* dirty restart streams: the one-lane emulation kernel only */
#define HJD_DBG_SIBLINGS(n) ((((n) + 1) & 15) << 8)  /* exactly n sibling trajectories (0..11) instead of the choice by launch size */
#define HJD_DBG_SEG_CUT(n) (((n) & 0x7FFF) << 16)    /* emulation: a segment is cut off after exactly n blocks (0 = default) */

hjd_decoder *hjd_decoder_create(int device, const hjd_options *opt /* may be NULL */);
void hjd_decoder_destroy(hjd_decoder *d);
int hjd_device_count(void);

/* Parse N JPEG files (host pointers), upload scans + tables to HBM, size the workspace.
 * status[i] receives the per-image parse status; images that fail to parse are skipped
 * by run/download.  Returns NULL only on argument / device errors. */
hjd_batch *hjd_batch_prepare(hjd_decoder *d, int n, const uint8_t *const *jpg, const size_t *jpg_len,
                             int out_format, int *status);
void hjd_batch_destroy(hjd_batch *b);
/* The same for images whose headers the caller has parsed — the batch-level twin of hjd_jpeg_decode_hw (loadjpg.h:180): jdata[i]
 * as that call takes it (width / height / factors inside the struct), validated like a parsed header.  restart_mcus may be 0 with
 * restart_interval > 0 (what an unmodified reference host knows, openjpg.cpp:441-446): the restart markers are then taken where
 * the scan holds them, as the reference does (loadjpg.cpp:535-550). */
hjd_batch *hjd_batch_prepare_parsed(hjd_decoder *d, int n, const hjd_jpeg_data *jdata, int out_format, int *status);

/* Enqueue the whole decode (all kernels) on `hip_stream` (a hipStream_t, may be NULL = default
 * stream).  Asynchronous; inputs and outputs stay in HBM.  Work the library puts on streams of its own (the decoder's three:
 * ranges of a split batch, small images' kernels) is joined to `hip_stream` before the call returns.  Outputs and statuses are
 * final after hjd_batch_finish: that call re-synchronises from the host where the device could not settle a state
 * (hjd_batch_resync_count) and runs the emulation of restart streams outside the clean-resync domain the FIRST time a batch's
 * contents need it (the later runs of the same contents carry the emulation launches themselves, on `hip_stream`).
 * Streams: the HIP runtime has four hardware queues per process (GPU_MAX_HW_QUEUES) and binds a stream to one at the stream's
 * first command; streams that share a queue run one behind the other.  Create the decoder — and use the streams you will pass
 * here — early in the process; INTEGRATION.md §9. */
int hjd_batch_run(hjd_batch *b, void *hip_stream);
/* Wait for the stream, verify on-device convergence flags, fill status[i] (may be NULL). */
int hjd_batch_finish(hjd_batch *b, void *hip_stream, int *status);

/* Per-image output location inside the batch's device slab. */
int hjd_batch_image_info(const hjd_batch *b, int i, unsigned *width, unsigned *height, size_t *out_offset,
                         size_t *out_bytes);
void *hjd_batch_out_device_ptr(const hjd_batch *b);
size_t hjd_batch_out_bytes(const hjd_batch *b);
/* Copy image i's output (RGB or BMP bytes) to host memory. */
int hjd_batch_download(hjd_batch *b, int i, uint8_t *dst, size_t dst_cap);

/* Algorithmic bytes of the batch: sum over decodable images of
 * scan_bytes + header_bytes + 3*W*H; and total pixels. */
int hjd_batch_totals(const hjd_batch *b, uint64_t *pixels, uint64_t *algorithmic_bytes, uint64_t *scan_bytes);

/* Per-kernel device time of ONE run measured with hipEvents on `hip_stream` (diagnostics /
 * bench.py roofline).  names[k] are static strings; returns the number of stages written. */
#define HJD_MAX_STAGES 16
int hjd_batch_run_timed(hjd_batch *b, void *hip_stream, float *stage_ms, const char **stage_names, int cap);

/* Stage timing inside the normal run path: after hjd_batch_enable_timing(b, R) the next R calls of
 * hjd_batch_run record a hipEvent before every stage on the run's stream (no extra synchronisation);
 * hjd_batch_collect_timing (call it after the stream has been synchronised) averages them. */
int hjd_batch_enable_timing(hjd_batch *b, int max_runs);
int hjd_batch_collect_timing(hjd_batch *b, float *stage_ms_avg, const char **stage_names, int cap, int *runs);

/* Diagnostics of the last finished run for image i: stats4 = {sum of refinement iterations over
 * synchronisation workgroups, subsequence decodes executed while synchronising, max iterations of one
 * workgroup, number of subsequences}. */
int hjd_batch_sync_stats(const hjd_batch *b, int i, uint32_t *stats4);
/* Diagnostics, only with HJD_SYNC_PROF=1 in the environment when the batch was laid out: shader-clock stamps of the stages of
 * every synchronisation workgroup of the last run (32 words per workgroup: clock, tag pairs; tools/sync_profile.py).  Returns the
 * number of workgroups, or -1. */
int hjd_debug_sync_profile(hjd_batch *b, unsigned long long *dst, size_t cap_words);
/* Which kernels the batch's images were laid out for: counts4 = {images synchronised by k_sync_seq (one lane per image), by
 * k_sync_small (one wave), by the large k_sync; images whose final pass runs in k_write_small}. */
int hjd_batch_path_counts(const hjd_batch *b, uint32_t *counts4);
/* Number of host-driven re-synchronisation rounds hjd_batch_finish had to run since prepare
 * (0 in normal operation: the warm-up subsequences of every synchronisation workgroup settled every state). */
int hjd_batch_resync_count(const hjd_batch *b);
/* Number of images of this batch that went through the emulation of the reference's sequential decoder (restart streams outside the
 * clean-resync domain) in the runs finished since prepare / reset.  The emulation is enqueued by hjd_batch_finish the first time
 * these contents need it and by hjd_batch_run afterwards: statuses and outputs of such images are final after hjd_batch_finish. */
int hjd_batch_sequential_count(const hjd_batch *b);
/* ... and how many of those were decoded in parallel over the places where the reference resynchronises (csrc/hjd_seq.hip: k_seg_*)
 * rather than by one lane per image. */
int hjd_batch_segment_count(const hjd_batch *b);

/* A batch object that can be prepared again and again in place: slabs and pinned staging are re-allocated only
 * when new contents need more than they hold.  hjd_batch_reset = the work of hjd_batch_prepare on an existing
 * object, with every upload enqueued as an asynchronous copy on `hip_stream` (the caller orders later work
 * after it; the caller's JPEG buffers are not referenced once the call returns). */
hjd_batch *hjd_batch_create(hjd_decoder *d);
int hjd_batch_reset(hjd_batch *b, int n, const uint8_t *const *jpg, const size_t *jpg_len, int out_format, int *status,
                    void *hip_stream);

/* Host JPEG bytes in, host RGB/BMP bytes out, for a whole batch: the file-in/file-out contract of the reference's
 * ConvertJpgFile (openjpg.cpp:593-684) without the files.  Pipelined: header walk + H2D of chunk k+1, kernels of
 * chunk k and D2H of chunk k-1 overlap on three streams.  out[i] receives image i (out_cap[i] >= its size);
 * buffers from hjd_host_alloc (pinned) are written by the DMA engine directly, pageable ones through a pinned
 * staging buffer and a host memcpy.  Returns when every output is complete. */
int hjd_decode_batch_host(hjd_decoder *d, int n, const uint8_t *const *jpg, const size_t *jpg_len,
                          int out_format, uint8_t *const *out, const size_t *out_cap, int *status);
void *hjd_host_alloc(size_t bytes); /* pinned host memory; NULL on failure */
void hjd_host_free(void *p);

/* Size helpers (WriteBMP24 arithmetic, openjpg.cpp:534-541). */
size_t hjd_bmp_bytes(unsigned width, unsigned height);
size_t hjd_rgb_bytes(unsigned width, unsigned height);

/* Test hooks: run single stages on device buffers so each kernel can be checked against the
 * oracle in isolation.  coefs: nblocks*64 int16, zig-zag order, DC predictor applied (the
 * DCT[] array after ProcessHuffmanBlock, loadjpg.cpp:499-863). */
int hjd_batch_download_coefs(hjd_batch *b, int i, int16_t *dst, size_t dst_elems);
/* The colour stage alone (ConvertYCrCbtoRGB + Clamp, loadjpg.cpp:867-880, :83-91) over all 2^24 inputs, run by
 * the device functions the decode kernel uses: dst[((Y << 16 | Cb << 8 | Cr) * 3) + {0,1,2}] = R, G, B.
 * dst_cap >= 3 << 24 bytes (host memory). */
int hjd_test_color_table(hjd_decoder *d, uint8_t *dst, size_t dst_cap);

/* Test hook, no device needed: the HBM / LDS layout hjd_batch_prepare would give ONE image with these header fields
 * and this scan length (only width, height, hfactor, vfactor, stream_len, restart fields and the table specs of
 * `j` matter; `stream` must be non-NULL but is not read).  tests/test_layout_bounds.py recomputes from it the
 * largest element every kernel may touch in every slab. */
typedef struct hjd_layout_info {
    uint32_t subseq_bits, sync_overlap, phase_words, sync_wg, write_wg, group_mcus, unstuff_chunk, dc_chunk, rec_run;
    uint32_t nblocks, mcux, mcuy, nb, nchunks, nsub_cap, nsyncwg, nrepwg, nwritewg, ndcchunk, ngroups, groups_per_row;
    uint32_t mark_cap, rec_cap, out_stride, lds_bytes_sync, lds_bytes_write, lds_state_word, lds_state_word_sync;
    uint64_t raw_bytes, ustream_bytes, blk_entries, dca_entries, rec_entries, out_bytes, substate_entries, wgsum_entries,
        dcsum_entries, ckpt_entries;
    /* segment-parallel emulation of dirty restart streams (csrc/hjd_seq.hip; all 0 without DRI): candidates reserved, nodes
     * (= (seg_cap + 1) * nb), bytes of the slab that holds the candidate list, the per-node arrays, counters and plans */
    uint64_t seg_cap, seg_nodes, seg_slab_bytes;
} hjd_layout_info;
int hjd_layout_query(const hjd_jpeg_data *j, int subseq_bits, int sync_overlap, int out_format, hjd_layout_info *out);

#ifdef __cplusplus
}
#endif
#endif /* HJD_H */
