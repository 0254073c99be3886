/*
 * hjd_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, single-threaded restatement of the baseline-JPEG -> RGB/BMP path of
 * harutel/hls-jpeg-decoder (reference files src/loadjpg.cpp, src/openjpg.cpp).
 * It exists only so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg can check the HIP path against it.  Nothing under
 * hls-jpeg-decoder_amd/ may include, link or call this file.
 *
 * Parity pinning: this restatement is checked (tests/test_oracle_pin.py) against
 *   - the reference's own fixture data/Lenna.jpg (BMP sha256 recorded in
 *     tests/golden/manifest.json, produced by the unmodified reference compiled
 *     by oracle/Makefile into oracle/_ref/),
 *   - golden BMP hashes of the committed synthetic JPEGs in tests/golden/,
 *     produced by that same compiled reference (script tools/make_golden.py).
 *
 * Every function cites the reference file:line whose behaviour it restates.
 */
#ifndef HJD_ORACLE_H
#define HJD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    HJO_OK = 0,
    HJO_ERR_NOT_JPEG = 1,      /* openjpg.cpp:481-486, :383-386 */
    HJO_ERR_UNSUPPORTED = 2,   /* outside the parity domain of SURVEY App. A.2 */
    HJO_ERR_CORRUPT = 3,       /* truncated / malformed segment */
    HJO_ERR_REF_WOULD_HANG = 4,/* loadjpg.cpp:700-829 no-progress loop */
    HJO_ERR_REF_OVERRUN = 5,   /* loadjpg.cpp:778-806 writes DCT[nr], nr > 63 */
    HJO_ERR_DC_CODE_MISSING = 6 /* loadjpg.cpp:680-685 */
};

/* One Huffman table as the reference keeps it: (length, code, value) triples in
 * DHT order (openjpg.cpp:48-98). */
typedef struct {
    int defined;
    int ncodes;
    uint8_t length[256];
    uint16_t code[256];
    uint8_t value[256];
} hjo_huff_t;

typedef struct {
    unsigned width, height;
    unsigned hf, vf;              /* luma sampling factors (openjpg.cpp:344-345) */
    int restart_enabled;          /* openjpg.cpp:441-446: only ever used as ">0" */
    unsigned restart_interval;    /* the DRI payload Ri; the reference never reads it */
    /* quantiser tables per component id 1..3, raw DQT (zig-zag) order, openjpg.cpp:347-350 */
    uint8_t qt[4][64];
    unsigned dc_sel[4], ac_sel[4];/* per component id, openjpg.cpp:212-213 */
    hjo_huff_t dc[4], ac[4];
    size_t scan_offset;           /* first entropy-coded byte, openjpg.cpp:220-223 */
    size_t scan_len;              /* bytes from scan_offset to end of buffer */
    size_t header_bytes;
} hjo_image_t;

typedef struct {
    size_t final_stream_index;    /* loadjpg.h:150 after the last block */
    unsigned resyncs_block_start; /* loadjpg.cpp:535-550 firings */
    unsigned resyncs_mid_block;   /* loadjpg.cpp:631-640 firings */
    unsigned resyncs_dirty;       /* firings (or misses) outside the clean-resync domain, SURVEY App. D.4 */
    unsigned long nblocks;
    unsigned long nonzero_coefs;
} hjo_stats_t;

/* Header walk: JpegParseHeader/ParseJFIF/ParseDQT/ParseSOF/ParseDHT/ParseSOS,
 * openjpg.cpp:120-496. */
int hjo_parse(const uint8_t *buf, size_t n, hjo_image_t *im);

/* Whole path: JpegDecodeHW, loadjpg.cpp:1134-1190.  rgb = 3*W*H bytes, top-down
 * R,G,B.  coefs (optional) = nblocks*64 int16 in zig-zag order with the DC
 * predictor applied (the DCT[] array after ProcessHuffmanBlock).  planes
 * (optional) = per-MCU u8 tiles in the m_Y/m_Cb/m_Cr layout, (hf*vf+2)*64 bytes
 * per MCU. */
int hjo_decode(const uint8_t *buf, size_t n, uint8_t *rgb, int16_t *coefs,
               uint8_t *planes, hjo_stats_t *stats);

/* DecodeSingleBlock, loadjpg.cpp:184-228 (dequant, de-zig-zag, transpose, IDCT,
 * +128, clamp).  coef_zz: 64 int16 zig-zag order; q_zz: 64 quantiser values in
 * zig-zag order; out: 8 rows of 8 at `stride`.  full_terms != 0 evaluates all
 * 64 terms (the literal loadjpg.cpp:105-124 loop); 0 skips zero coefficients. */
void hjo_block_idct(const int16_t *coef_zz, const uint8_t *q_zz, uint8_t *out,
                    int stride, int full_terms);

/* ConvertYCrCbtoRGB as called at loadjpg.cpp:918 (y, cb, cr are tile values). */
void hjo_ycc_to_rgb(int y, int cb, int cr, uint8_t *r, uint8_t *g, uint8_t *b);

/* WriteBMP24 byte image, openjpg.cpp:504-570. */
size_t hjo_bmp_size(unsigned w, unsigned h);
void hjo_bmp_pack(const uint8_t *rgb, unsigned w, unsigned h, uint8_t *out);

/* ConvertJpgFile, openjpg.cpp:593-684: returns 1 on success, 0 on failure. */
int hjo_convert_jpg_file(const char *jpg, const char *bmp);

/* cosine table COS[x][u] */
const float *hjo_cos_table(void);

#ifdef __cplusplus
}
#endif
#endif
