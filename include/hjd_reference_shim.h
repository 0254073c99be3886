/*
 * hjd_reference_shim.h — the binding a maintainer of harutel/hls-jpeg-decoder adds to keep the reference's own
 * header parser (JpegParseHeader, openjpg.cpp:478) and replace only the HLS top function (JpegDecodeHW,
 * loadjpg.h:180) by libhjd.so.  Integration glue, header-only; it needs the REFERENCE's headers (loadjpg.h,
 * openjpg.h) on the include path and is therefore not compiled into the library.  INTEGRATION.md §2 shows its
 * use; tests/test_reference_shim.py compiles it against the reference's sources (build container only) and
 * compares what it produces with hjd_parse_header.
 */
#ifndef HJD_REFERENCE_SHIM_H
#define HJD_REFERENCE_SHIM_H

#include <string.h>

#include "hjd.h"
#include "openjpg.h" /* the reference's: stJpegData, stImageInfo, cY/cCb/cCr, STREAM_SIZE */

/* Length of the entropy-coded segment that starts at `s`: up to and including the EOI marker (FF D9), at most
 * `cap` bytes.  ParseSOS copies a fixed STREAM_SIZE bytes whatever the file holds (openjpg.cpp:220-223), so
 * m_stream runs on into stale bytes after EOI; a stuffed FF 00 and the RSTn markers are part of the scan. */
static inline size_t hjd_shim_scan_length(const unsigned char *s, size_t cap)
{
    for (size_t i = 0; i + 1 < cap; i++)
        if (s[i] == 0xFF && s[i + 1] == 0xD9) return i + 2;
    return cap;
}

/* stJpegData (loadjpg.h:157-171) + stImageInfo (openjpg.h:11-17) -> hjd_jpeg_data: pointers and lengths instead of
 * fixed-capacity arrays.  dri_mcus = the DRI payload Ri, which the reference never stores (openjpg.cpp:441-446
 * keeps the segment LENGTH); pass 0 when the file has no DRI. */
static inline void hjd_shim_from_reference(const stJpegData *jd, const stImageInfo *ji, unsigned dri_mcus, hjd_jpeg_data *o)
{
    memset(o, 0, sizeof *o);
    for (int c = cY; c <= cCr; c++) {
        for (int k = 0; k < 64; k++) o->qtable[c][k] = (uint8_t)jd->m_component_info[c].m_qTable[k]; /* openjpg.cpp:347-350 */
        o->dc_table_index[c] = (uint8_t)jd->m_component_info[c].dcTable_index;                       /* openjpg.cpp:213 */
        o->ac_table_index[c] = (uint8_t)jd->m_component_info[c].acTable_index;                       /* openjpg.cpp:212 */
    }
    for (int t = 0; t < 4; t++) {
        const stHuffmanTable *src[2] = {&jd->m_Huffman.m_HTDC[t], &jd->m_Huffman.m_HTAC[t]};
        hjd_huff_spec *dst[2] = {&o->htdc[t], &o->htac[t]};
        for (int k = 0; k < 2; k++) { /* (length, code, value) triples in code order, openjpg.cpp:48-98 */
            const int n = src[k]->m_numBlocks;
            if (n <= 0 || n > 256) continue;
            for (int i = 0; i < n; i++) {
                const int len = src[k]->m_blocks[i].length;
                if (len >= 1 && len <= 16) dst[k]->bits[len]++;
                dst[k]->vals[i] = (uint8_t)src[k]->m_blocks[i].value;
            }
            dst[k]->defined = 1;
        }
    }
    o->stream = jd->m_Huffman.m_stream; /* openjpg.cpp:220-223 */
    o->stream_len = hjd_shim_scan_length(jd->m_Huffman.m_stream, STREAM_SIZE);
    o->restart_interval = jd->m_Huffman.m_restart_interval; /* tested "> 0" only (loadjpg.cpp:535) */
    o->restart_mcus = dri_mcus;
    o->width = ji->m_width;
    o->height = ji->m_height;
    o->hfactor = ji->m_hFactor[cY];
    o->vfactor = ji->m_vFactor[cY];
}

/* Drop-in body for the reference's JpegDecodeHW: same arguments, same 0 on success (the reference returns 0
 * always; here a refused or corrupt input gives -1 and hjd_last_status() says why). */
static inline int hjd_shim_jpeg_decode_hw(stJpegData *jdata, const stImageInfo *ji, unsigned dri_mcus, unsigned int h, unsigned int w,
                                          unsigned char hF, unsigned char vF)
{
    hjd_jpeg_data d;
    hjd_shim_from_reference(jdata, ji, dri_mcus, &d);
    return hjd_jpeg_decode_hw(&d, h, w, hF, vF, jdata->m_rgb) == HJD_OK ? 0 : -1; /* same argument order */
}

#endif /* HJD_REFERENCE_SHIM_H */
