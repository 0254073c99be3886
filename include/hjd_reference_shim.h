/*
 * hjd_reference_shim.h — the binding a maintainer of harutel/hls-jpeg-decoder adds to keep the reference's own
 * header parser (JpegParseHeader, openjpg.cpp:478) and replace only the HLS top function (JpegDecodeHW,
 * loadjpg.h:180) by libhjd.so.  Integration glue, header-only; it needs the REFERENCE's headers (loadjpg.h,
 * openjpg.h) on the include path and is therefore not compiled into the library.  INTEGRATION.md §2 shows its
 * use; the test recipe builds the reference's unchanged host around it (refhost_hjd, run on the GPU by
 * tests/test_gpu_parity.py), and tests/test_reference_shim.py compares what it hands over with hjd_parse_header.
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

/* stJpegData (loadjpg.h:157-171) -> hjd_jpeg_data: pointers and lengths instead of fixed-capacity arrays.  Width, height and the
 * luma factors are the ones JpegDecodeHW is called with (openjpg.cpp:655): the struct does not hold them.  dri_mcus = the DRI
 * payload Ri, which the reference never stores (openjpg.cpp:441-446 keeps the segment LENGTH): a host that has not been changed
 * passes 0, and the library then takes the restart markers where the scan holds them — as the reference does
 * (loadjpg.cpp:535-550). */
static inline void hjd_shim_from_reference_args(const stJpegData *jd, unsigned int height, unsigned int width, unsigned char hF,
                                                unsigned char vF, unsigned dri_mcus, hjd_jpeg_data *o)
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
    o->width = width;
    o->height = height;
    o->hfactor = hF;
    o->vfactor = vF;
}

/* The same from the parser's two structs (stImageInfo, openjpg.h:11-17), for hosts that have them at hand. */
static inline void hjd_shim_from_reference(const stJpegData *jd, const stImageInfo *ji, unsigned dri_mcus, hjd_jpeg_data *o)
{
    hjd_shim_from_reference_args(jd, ji->m_height, ji->m_width, ji->m_hFactor[cY], ji->m_vFactor[cY], dri_mcus, o);
}

/* Drop-in BODY for the reference's JpegDecodeHW (loadjpg.h:180): same arguments, nothing else needed — host/reference_decodehw.cpp
 * is exactly `int JpegDecodeHW(...) { return hjd_shim_jpeg_decode_hw(...); }`, linked with the reference's unchanged main.cpp and
 * openjpg.cpp in place of loadjpg.cpp.  Returns 0 on success like the reference (which returns 0 always); a refused or corrupt
 * input gives -1 and hjd_last_status() says why. */
static inline int hjd_shim_jpeg_decode_hw(stJpegData *jdata, unsigned int jpeg_img_height, unsigned int jpeg_img_width,
                                          unsigned char hFactor, unsigned char vFactor)
{
    hjd_jpeg_data d;
    hjd_shim_from_reference_args(jdata, jpeg_img_height, jpeg_img_width, hFactor, vFactor, 0u, &d);
    return hjd_jpeg_decode_hw(&d, jpeg_img_height, jpeg_img_width, hFactor, vFactor, jdata->m_rgb) == HJD_OK ? 0 : -1; /* same argument order */
}

/* ... and for a host that was taught to keep Ri (one line at openjpg.cpp:441-446) */
static inline int hjd_shim_jpeg_decode_hw_ri(stJpegData *jdata, unsigned dri_mcus, unsigned int h, unsigned int w, unsigned char hF,
                                             unsigned char vF)
{
    hjd_jpeg_data d;
    hjd_shim_from_reference_args(jdata, h, w, hF, vF, dri_mcus, &d);
    return hjd_jpeg_decode_hw(&d, h, w, hF, vF, jdata->m_rgb) == HJD_OK ? 0 : -1;
}

#endif /* HJD_REFERENCE_SHIM_H */
