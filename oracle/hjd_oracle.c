/*
 * hjd_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 * See hjd_oracle.h for what this is, who may call it and how it is pinned.
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction: the reference's x86-64
 * build evaluates every fp32 operation with its own rounding).
 */
#include "hjd_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- constants ---------------------------------------------------------- */

/* natural index -> zig-zag index, loadjpg.cpp:56-66 */
static const uint8_t ZZ[64] = {
    0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42,
    3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
    10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
    21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

/* COS[x][u] = cosf(fl((2x+1)*u*3.14f)/16) as glibc 2.35 evaluates it on x86-64
 * (loadjpg.cpp:108,120).  Hard-coded: entry [6][7] is libm-specific (SURVEY
 * App. B); tools/gen_costable.c regenerates it from the reference's expression with the local libm and compares. */
static const float COS[8][8] = {
    {0x1p+0f, 0x1.f62c08p-1f, 0x1.d910b8p-1f, 0x1.a9cc2p-1f, 0x1.6a2ecap-1f, 0x1.1ca9eep-1f, 0x1.886ecap-2f, 0x1.90f1bep-3f},
    {0x1p+0f, 0x1.a9cc2p-1f, 0x1.886ecap-2f, -0x1.8dbedp-3f, -0x1.699b2p-1f, -0x1.f6032cp-1f, -0x1.d9606ap-1f, -0x1.1d5756p-1f},
    {0x1p+0f, 0x1.1ca9eep-1f, -0x1.86ed1p-2f, -0x1.f6032cp-1f, -0x1.6ac236p-1f, 0x1.8a8b92p-3f, 0x1.d8706ep-1f, 0x1.aab36ap-1f},
    {0x1p+0f, 0x1.90f1bep-3f, -0x1.d8c0bap-1f, -0x1.1d5756p-1f, 0x1.690742p-1f, 0x1.aab36ap-1f, -0x1.83e8d8p-2f, -0x1.f6a4aep-1f},
    {0x1p+0f, -0x1.8dbedp-3f, -0x1.d9606ap-1f, 0x1.1b4e94p-1f, 0x1.6b556cp-1f, -0x1.a86f2p-1f, -0x1.8cf256p-2f, 0x1.f5869ep-1f},
    {0x1p+0f, -0x1.1bfc56p-1f, -0x1.89f044p-2f, 0x1.f67ccap-1f, -0x1.68731cp-1f, -0x1.9a88c4p-3f, 0x1.da4da2p-1f, -0x1.a7851ep-1f},
    {0x1p+0f, -0x1.a9581p-1f, 0x1.856b06p-2f, 0x1.9756cp-3f, -0x1.6be866p-1f, 0x1.f6cc3cp-1f, -0x1.d77dbp-1f, 0x1.1895aep-1f},
    {0x1p+0f, -0x1.f6032cp-1f, 0x1.d8706ep-1f, -0x1.a86f2p-1f, 0x1.67dec4p-1f, -0x1.19442cp-1f, 0x1.7f609p-2f, -0x1.7a87f6p-3f}};

const float *hjo_cos_table(void) { return &COS[0][0]; }

/* C(u), loadjpg.cpp:96-102: 1.0f/sqrtf(2) = 0x1.6a09e6p-1 */
static inline float Cfac(int u) { return u == 0 ? 0x1.6a09e6p-1f : 1.0f; }

static inline uint8_t clamp255(int i) /* loadjpg.cpp:83-91 */
{
    return i < 0 ? 0 : (i > 255 ? 255 : (uint8_t)i);
}

/* ---- header walk -------------------------------------------------------- */

#define WORD(p) (((unsigned)(p)[0] << 8) | (unsigned)(p)[1]) /* loadjpg.h:79 */

/* GenHuffCodes + BuildHuffmanTable, openjpg.cpp:48-98 */
static int build_huff(hjo_huff_t *t, const uint8_t *bits /*[1..16]*/, const uint8_t *vals)
{
    int n = 0;
    for (int i = 1; i <= 16; i++)
        for (int j = 0; j < bits[i]; j++) t->length[n++] = (uint8_t)i;
    t->ncodes = n;
    unsigned code = 0;
    int curlen = 1;
    for (int c = 0; c < n; c++) {
        while (t->length[c] > curlen) {
            code <<= 1;
            curlen++;
        }
        if (code >= (1u << t->length[c])) return HJO_ERR_CORRUPT; /* over-subscribed */
        t->code[c] = (uint16_t)code;
        t->value[c] = vals[c];
        code++;
    }
    t->defined = 1;
    return HJO_OK;
}

int hjo_parse(const uint8_t *buf, size_t n, hjo_image_t *im)
{
    memset(im, 0, sizeof(*im));
    if (n < 4 || buf[0] != 0xFF || buf[1] != 0xD8) return HJO_ERR_NOT_JPEG; /* openjpg.cpp:481 */

    uint8_t qtab[4][64];
    int qdef[4] = {0, 0, 0, 0};
    int sof_seen = 0;
    int comp_seen[4] = {0, 0, 0, 0};
    unsigned comp_h[4] = {0}, comp_v[4] = {0};
    size_t p = 2;

    for (;;) { /* ParseJFIF marker loop, openjpg.cpp:381-462 */
        if (p >= n) return HJO_ERR_CORRUPT;
        if (buf[p++] != 0xFF) return HJO_ERR_NOT_JPEG; /* :383-386 */
        while (p < n && buf[p] == 0xFF) p++;            /* :389-392 */
        if (p + 3 > n) return HJO_ERR_CORRUPT;
        unsigned marker = buf[p++];
        const uint8_t *s = buf + p; /* at the length field */
        size_t avail = n - p;
        unsigned len = WORD(s);

        if (marker == 0xD8 || marker == 0xD9) { /* :433-439 */
            len = 0;
        } else if (marker == 0xDB) { /* ParseDQT :120-155 */
            if (len < 2 || len > avail) return HJO_ERR_CORRUPT;
            int rem = (int)len - 2;
            const uint8_t *q = s + 2;
            while (rem > 0) {
                if (rem < 65) return HJO_ERR_CORRUPT;
                unsigned qi = *q++;
                if (qi >> 4) return HJO_ERR_UNSUPPORTED;       /* :136-143 16-bit DQT */
                if ((qi & 0xf) >= 4) return HJO_ERR_UNSUPPORTED; /* :145-148 */
                memcpy(qtab[qi & 0xf], q, 64);                  /* :102-116 zig-zag order kept */
                qdef[qi & 0xf] = 1;
                q += 64;
                rem -= 65;
            }
        } else if (marker == 0xC0) { /* ParseSOF :310-367 */
            if (len < 8 || len > avail) return HJO_ERR_CORRUPT;
            if (s[2] != 8) return HJO_ERR_UNSUPPORTED;
            im->height = WORD(s + 3);
            im->width = WORD(s + 5);
            unsigned nc = s[7];
            if (nc != 3) return HJO_ERR_UNSUPPORTED;
            if (len < 8 + 3 * nc) return HJO_ERR_CORRUPT;
            const uint8_t *c = s + 8;
            for (unsigned i = 0; i < nc; i++, c += 3) {
                unsigned cid = c[0], samp = c[1], tq = c[2];
                if (cid < 1 || cid > 3) return HJO_ERR_UNSUPPORTED; /* :343 indexes by id */
                if (tq >= 4 || !qdef[tq]) return HJO_ERR_UNSUPPORTED; /* :347-350 copy at SOF time */
                comp_v[cid] = samp & 0xf; /* :344 */
                comp_h[cid] = samp >> 4;  /* :345 */
                memcpy(im->qt[cid], qtab[tq], 64);
                comp_seen[cid] = 1;
            }
            sof_seen = 1;
        } else if (marker == 0xC4) { /* ParseDHT :234-305 */
            if (len < 2 || len > avail) return HJO_ERR_CORRUPT;
            int rem = (int)len - 2;
            const uint8_t *h = s + 2;
            while (rem > 0) {
                if (rem < 17) return HJO_ERR_CORRUPT;
                unsigned index = *h++;
                uint8_t bits[17];
                unsigned count = 0;
                bits[0] = 0;
                for (int i = 1; i < 17; i++) {
                    bits[i] = *h++;
                    count += bits[i];
                }
                if (count > 255) return HJO_ERR_UNSUPPORTED; /* loadjpg.cpp:350,355 uint8_t loop index */
                if ((index & 0xf) >= 4) return HJO_ERR_UNSUPPORTED; /* :273-276 */
                if (rem < (int)(17 + count)) return HJO_ERR_CORRUPT;
                hjo_huff_t *t = (index & 0xf0) ? &im->ac[index & 0xf] : &im->dc[index & 0xf];
                int rc = build_huff(t, bits, h);
                if (rc) return rc;
                if (!(index & 0xf0)) {
                    for (int c = 0; c < t->ncodes; c++) {
                        if (t->length[c] > 15) return HJO_ERR_UNSUPPORTED; /* loadjpg.cpp:562 k<16 */
                        if (t->value[c] > 15) return HJO_ERR_UNSUPPORTED;  /* FillNBits loads <=2 bytes */
                    }
                }
                h += count;
                rem -= 17 + (int)count;
            }
        } else if (marker == 0xDD) { /* :441-446: stores the LENGTH field */
            if (len > avail) return HJO_ERR_CORRUPT;
            im->restart_enabled = len > 0;
            im->restart_interval = len >= 4 ? WORD(s + 2) : 0;
        } else if (marker == 0xDA) { /* ParseSOS :160-229 */
            if (!sof_seen) return HJO_ERR_UNSUPPORTED;
            if (avail < 3) return HJO_ERR_CORRUPT;
            unsigned ns = s[2];
            if (ns != 3) return HJO_ERR_UNSUPPORTED; /* :180-183 */
            if (avail < 3 + 2 * ns + 3) return HJO_ERR_CORRUPT;
            const uint8_t *c = s + 3;
            for (unsigned i = 0; i < ns; i++, c += 2) {
                unsigned cid = c[0], tab = c[1];
                if (cid != i + 1) return HJO_ERR_UNSUPPORTED; /* decode order is fixed Y,Cb,Cr (loadjpg.cpp:945-997) */
                if ((tab & 0xf) >= 4 || (tab >> 4) >= 4) return HJO_ERR_UNSUPPORTED;
                im->ac_sel[cid] = tab & 0xf; /* :212 */
                im->dc_sel[cid] = tab >> 4;  /* :213 */
                if (!im->ac[tab & 0xf].defined || !im->dc[tab >> 4].defined) return HJO_ERR_UNSUPPORTED;
            }
            im->scan_offset = p + 3 + 2 * ns + 3; /* :186-190, :222 "stream[i+3]" */
            im->scan_len = n - im->scan_offset;
            im->header_bytes = im->scan_offset;
            break;
        } else {
            /* APP0 / unknown: skipped by length, :448-458 */
            if (len > avail) return HJO_ERR_CORRUPT;
        }
        p += len; /* :461 */
    }

    if (!comp_seen[1] || !comp_seen[2] || !comp_seen[3]) return HJO_ERR_UNSUPPORTED;
    im->hf = comp_h[1];
    im->vf = comp_v[1];
    /* loadjpg.cpp:945-997 decodes exactly one Cb and one Cr block per MCU; m_Y holds
     * 4 blocks (loadjpg.h:165); loadjpg.cpp:912 index math is exact for 1,2,4 only */
    if (comp_h[2] != 1 || comp_v[2] != 1 || comp_h[3] != 1 || comp_v[3] != 1) return HJO_ERR_UNSUPPORTED;
    if (!(im->hf == 1 || im->hf == 2 || im->hf == 4)) return HJO_ERR_UNSUPPORTED;
    if (!(im->vf == 1 || im->vf == 2 || im->vf == 4)) return HJO_ERR_UNSUPPORTED;
    if (im->hf * im->vf > 4) return HJO_ERR_UNSUPPORTED;
    if (im->width == 0 || im->height == 0) return HJO_ERR_UNSUPPORTED;
    return HJO_OK;
}

/* ---- bit reader --------------------------------------------------------- */

typedef struct {
    const uint8_t *s; /* m_stream */
    size_t n;
    size_t idx;     /* stream_index, loadjpg.h:150 */
    uint32_t res;   /* g_reservoir */
    uint32_t nbits; /* g_nbits_in_reservoir */
} rd_t;

static inline uint8_t rd_at(const rd_t *r, size_t i) { return i < r->n ? r->s[i] : 0; }

/* FillNBits live branch, loadjpg.cpp:468-482 */
static inline void fill_nbits(rd_t *r, int limit)
{
    for (int k = 0; k < 2; k++) {
        if (r->nbits < (unsigned)limit) {
            uint8_t c = rd_at(r, r->idx);
            r->idx++;
            r->res <<= 8;
            if (c == 0xff && rd_at(r, r->idx) == 0x00) r->idx++;
            r->res |= c;
            r->nbits += 8;
        }
    }
}

static inline void skip_nbits(rd_t *r, int k) /* loadjpg.cpp:607-609 */
{
    fill_nbits(r, k);
    r->nbits -= (unsigned)k;
    r->res &= ((1u << r->nbits) - 1u);
}

/* canonical-code match for (code, length): same answer as the full-table scan of
 * IsInHuffmanCodes (loadjpg.cpp:335-392) for the prefix-free tables hjo_parse admits */
typedef struct {
    int first[17];  /* index of first code of this length, -1 if none */
    int count[17];
    unsigned mincode[17];
    const hjo_huff_t *t;
} hlook_t;

static void hlook_init(hlook_t *h, const hjo_huff_t *t)
{
    h->t = t;
    for (int k = 0; k < 17; k++) {
        h->first[k] = -1;
        h->count[k] = 0;
        h->mincode[k] = 0;
    }
    for (int c = 0; c < t->ncodes; c++) {
        int k = t->length[c];
        if (h->first[k] < 0) {
            h->first[k] = c;
            h->mincode[k] = t->code[c];
        }
        h->count[k]++;
    }
}

static inline int hlook_match(const hlook_t *h, unsigned code, int k, int *value)
{
    if (h->first[k] < 0) return 0;
    if (code < h->mincode[k] || code >= h->mincode[k] + (unsigned)h->count[k]) return 0;
    *value = h->t->value[h->first[k] + (int)(code - h->mincode[k])];
    return 1;
}

/* DetermineSign, loadjpg.cpp:396-409 */
static inline int determine_sign(int val, int nbits)
{
    if (val < (1 << (nbits - 1))) val = val + (int)(0xFFFFFFFFu << nbits) + 1;
    return val;
}

/* ProcessHuffmanBlock, loadjpg.cpp:499-863 */
static int huff_block(rd_t *r, int restart, int expect_resync, const hlook_t *hdc, const hlook_t *hac,
                      int16_t dct[64], int16_t *prev_dc, hjo_stats_t *st)
{
    memset(dct, 0, 64 * sizeof(int16_t)); /* :523-527 */
    int decoded = 0;
    int found = 0;

    if (restart && rd_at(r, r->idx) == 0xff && rd_at(r, r->idx + 1) != 0x00) { /* :535-550 */
        r->idx += 2;
        r->res = 0;
        r->nbits = 0;
        if (st) {
            st->resyncs_block_start++;
            if (!expect_resync) st->resyncs_dirty++;
        }
    } else if (restart && expect_resync && st) {
        st->resyncs_dirty++; /* marker expected here but the byte cursor is not on it */
    }

    for (int k = 1; k < 16; k++) { /* :562 */
        fill_nbits(r, k);
        unsigned code = r->res >> (r->nbits - (unsigned)k); /* :584 */
        if (hlook_match(hdc, code, k, &decoded)) {
            skip_nbits(r, k);
            found = 1;
            int nbits = decoded;
            if (nbits == 0) {
                dct[0] = *prev_dc; /* :625 */
            } else {
                if (restart && rd_at(r, r->idx) == 0xff && rd_at(r, r->idx + 1) != 0x00) { /* :631-640 */
                    r->idx += 2;
                    r->res = 0;
                    r->nbits = 0;
                    if (st) {
                        st->resyncs_mid_block++;
                        st->resyncs_dirty++;
                    }
                }
                fill_nbits(r, nbits);
                int16_t data = (int16_t)(r->res >> (r->nbits - (unsigned)nbits)); /* :649 */
                r->nbits -= (unsigned)nbits;
                r->res &= ((1u << r->nbits) - 1u);
                data = (int16_t)determine_sign(data, nbits);
                dct[0] = (int16_t)(data + *prev_dc); /* :664 */
                *prev_dc = dct[0];
            }
            break;
        }
    }
    if (!found) return HJO_ERR_DC_CODE_MISSING; /* :680-685 (reference prints and carries on) */

    int nr = 1;
    int eob = 0;
    while (nr <= 63 && !eob) { /* :700 */
        int k, hit = 0;
        for (k = 1; k <= 16; k++) { /* :713 */
            fill_nbits(r, k);
            unsigned code = r->res >> (r->nbits - (unsigned)k);
            if (hlook_match(hac, code, k, &decoded)) {
                skip_nbits(r, k);
                hit = 1;
                unsigned size_val = (unsigned)decoded & 0xF; /* :768 */
                unsigned count_0 = (unsigned)decoded >> 4;   /* :769 */
                if (size_val == 0) {
                    if (count_0 == 0) eob = 1;           /* :773 */
                    else if (count_0 == 0xF) nr += 16;   /* :774 */
                    /* other run lengths with size 0: ignored */
                } else {
                    nr += (int)count_0; /* :778 */
                    if (nr > 63) return HJO_ERR_REF_OVERRUN; /* :780-806 */
                    fill_nbits(r, (int)size_val);
                    int16_t data = (int16_t)(r->res >> (r->nbits - size_val)); /* :795 */
                    r->nbits -= size_val;
                    r->res &= ((1u << r->nbits) - 1u);
                    data = (int16_t)determine_sign(data, (int)size_val);
                    dct[nr++] = data; /* :806 */
                    if (st) st->nonzero_coefs++;
                }
                break;
            }
        }
        if (!hit) return HJO_ERR_REF_WOULD_HANG; /* no code of length 1..16: loop makes no progress */
    }
    return HJO_OK;
}

/* ---- block back end ----------------------------------------------------- */

void hjo_block_idct(const int16_t *coef_zz, const uint8_t *q_zz, uint8_t *out, int stride,
                    int full_terms)
{
    /* DequantizeBlock :144-152 (float product is exact, then short wrap),
     * DeZigZag :156-163, TransformArray :167-180: arr[u][v] = nat[v*8+u] */
    int16_t arr[8][8];
    for (int i = 0; i < 64; i++) {
        int prod = (int)((float)coef_zz[ZZ[i]] * (float)q_zz[ZZ[i]]);
        int16_t nat = (int16_t)prod;
        arr[i & 7][i >> 3] = nat;
    }

    float sum[8][8]; /* [x][y] */
    for (int x = 0; x < 8; x++)
        for (int y = 0; y < 8; y++) sum[x][y] = 0.0f;

    /* IDCT_calc :105-124: u outer, v inner; term = (((C(u)*C(v))*b)*cos_x)*cos_y */
    for (int u = 0; u < 8; u++) {
        for (int v = 0; v < 8; v++) {
            int16_t b = arr[u][v];
            if (!full_terms && b == 0) continue; /* adds +-0: no effect */
            float k = Cfac(u) * Cfac(v);
            float t1 = k * (float)b;
            for (int x = 0; x < 8; x++) {
                float t2 = t1 * COS[x][u];
                for (int y = 0; y < 8; y++) {
                    float t3 = t2 * COS[y][v];
                    sum[x][y] = sum[x][y] + t3;
                }
            }
        }
    }
    for (int y = 0; y < 8; y++) { /* PerformIDCT :126-140, store loop :213-223 */
        for (int x = 0; x < 8; x++) {
            int16_t val = (int16_t)(int)(0.25 * (double)sum[x][y]); /* :123 */
            val = (int16_t)(val + 128);                             /* :137 */
            out[y * stride + x] = clamp255(val);
        }
    }
}

void hjo_ycc_to_rgb(int y, int cb, int cr, uint8_t *r, uint8_t *g, uint8_t *b)
{
    /* loadjpg.cpp:867-880 with the argument swap of :918 resolved */
    float red = (float)y + 1.402f * (float)(cr - 128);
    float green = (float)y - 0.34414f * (float)(cb - 128) - 0.71414f * (float)(cr - 128);
    float blue = (float)y + 1.772f * (float)(cb - 128);
    *r = clamp255((int)red);
    *g = clamp255((int)green);
    *b = clamp255((int)blue);
}

/* ---- whole path --------------------------------------------------------- */

int hjo_decode(const uint8_t *buf, size_t n, uint8_t *rgb, int16_t *coefs, uint8_t *planes,
               hjo_stats_t *stats)
{
    hjo_image_t *im = (hjo_image_t *)malloc(sizeof(hjo_image_t));
    if (!im) return HJO_ERR_CORRUPT;
    int rc = hjo_parse(buf, n, im);
    if (rc) {
        free(im);
        return rc;
    }
    hjo_stats_t st_local;
    hjo_stats_t *st = stats ? stats : &st_local;
    memset(st, 0, sizeof(*st));

    const unsigned W = im->width, H = im->height, hf = im->hf, vf = im->vf;
    hlook_t hdc[4], hac[4];
    for (int c = 1; c <= 3; c++) {
        hlook_init(&hdc[c], &im->dc[im->dc_sel[c]]);
        hlook_init(&hac[c], &im->ac[im->ac_sel[c]]);
    }

    rd_t r = {buf + im->scan_offset, im->scan_len, 0, 0, 0}; /* loadjpg.cpp:1148-1149 */
    int16_t prev[4] = {0, 0, 0, 0};                            /* :1159-1162 */
    int16_t dct[64];
    uint8_t tY[256], tCb[64], tCr[64];
    const unsigned blocks_per_mcu = hf * vf + 2;
    unsigned long mcu = 0;

    for (unsigned y0 = 0; y0 < H && rc == HJO_OK; y0 += 8 * vf) {     /* :1170 */
        for (unsigned x0 = 0; x0 < W && rc == HJO_OK; x0 += 8 * hf, mcu++) { /* :1174 */
            /* DecodeMCU :945-997 */
            unsigned bi = 0;
            const int expect = im->restart_enabled && im->restart_interval && mcu && (mcu % im->restart_interval) == 0;
            for (unsigned by = 0; by < vf && rc == HJO_OK; by++) {
                for (unsigned bx = 0; bx < hf && rc == HJO_OK; bx++, bi++) {
                    rc = huff_block(&r, im->restart_enabled, expect && bi == 0, &hdc[1], &hac[1], dct, &prev[1], st);
                    if (rc) break;
                    st->nblocks++;
                    if (coefs) memcpy(coefs + (mcu * blocks_per_mcu + bi) * 64, dct, 128);
                    hjo_block_idct(dct, im->qt[1], tY + bx * 8 + by * 64 * hf, (int)(hf * 8), 0); /* :958-970 */
                }
            }
            if (rc) break;
            rc = huff_block(&r, im->restart_enabled, 0, &hdc[2], &hac[2], dct, &prev[2], st);
            if (rc) break;
            st->nblocks++;
            if (coefs) memcpy(coefs + (mcu * blocks_per_mcu + bi) * 64, dct, 128);
            hjo_block_idct(dct, im->qt[3], tCb, 8, 0); /* :984 — Cb is dequantised with Cr's table */
            bi++;
            rc = huff_block(&r, im->restart_enabled, 0, &hdc[3], &hac[3], dct, &prev[3], st);
            if (rc) break;
            st->nblocks++;
            if (coefs) memcpy(coefs + (mcu * blocks_per_mcu + bi) * 64, dct, 128);
            hjo_block_idct(dct, im->qt[3], tCr, 8, 0); /* :996 */

            if (planes) {
                uint8_t *pl = planes + mcu * blocks_per_mcu * 64;
                memcpy(pl, tY, hf * vf * 64);
                memcpy(pl + hf * vf * 64, tCb, 64);
                memcpy(pl + hf * vf * 64 + 64, tCr, 64);
            }

            /* YCrCB_to_RGB24_Block8x8 :884-932 */
            if (rgb) {
                for (unsigned y = 0; y < 8 * vf; y++) {
                    if (y + y0 >= H) continue; /* :908 */
                    for (unsigned x = 0; x < 8 * hf; x++) {
                        if (x + x0 >= W) continue; /* :907 */
                        unsigned yoff = x + y * (hf * 8);
                        unsigned coff = (x / hf) + (y / vf) * 8; /* :912, exact for 1,2,4 */
                        uint8_t *px = rgb + ((size_t)(y0 + y) * W + (x0 + x)) * 3;
                        hjo_ycc_to_rgb(tY[yoff], tCb[coff], tCr[coff], px, px + 1, px + 2);
                    }
                }
            }
        }
    }
    st->final_stream_index = r.idx;
    free(im);
    return rc;
}

/* ---- BMP ---------------------------------------------------------------- */

static unsigned bmp_pad(unsigned w) { return (4 - (w * 3) % 4) % 4; } /* openjpg.cpp:534-535 */

size_t hjo_bmp_size(unsigned w, unsigned h)
{
    return (size_t)w * h * 3 + (size_t)h * bmp_pad(w) + 54; /* :541 */
}

static void put32(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
    p[2] = (uint8_t)(v >> 16);
    p[3] = (uint8_t)(v >> 24);
}

void hjo_bmp_pack(const uint8_t *rgb, unsigned w, unsigned h, uint8_t *out)
{
    unsigned pad = bmp_pad(w);
    memset(out, 0, 54); /* :538 */
    out[0] = 'B';
    out[1] = 'M';
    put32(out + 2, (uint32_t)(w * h * 3 + h * pad + 54)); /* :541 (32-bit arithmetic) */
    put32(out + 10, 54);
    put32(out + 14, 40);
    put32(out + 18, w);
    put32(out + 22, h);
    out[26] = 1;  /* planes */
    out[28] = 24; /* bit count */
    uint8_t *o = out + 54;
    for (int y = (int)h - 1; y >= 0; y--) { /* :555 bottom-up */
        for (unsigned x = 0; x < w; x++) {
            const uint8_t *px = rgb + ((size_t)y * w + x) * 3;
            *o++ = px[2]; /* :560-561 (R<<16|G<<8|B) little-endian, 3 bytes: B,G,R */
            *o++ = px[1];
            *o++ = px[0];
        }
        for (unsigned k = 0; k < pad; k++) *o++ = 0; /* :563-567 */
    }
}

int hjo_convert_jpg_file(const char *jpg, const char *bmp)
{
    FILE *fp = fopen(jpg, "rb");
    if (!fp) return 0; /* openjpg.cpp:604-608 */
    fseek(fp, 0, SEEK_END);
    long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (sz <= 0) {
        fclose(fp);
        return 0;
    }
    uint8_t *buf = (uint8_t *)malloc((size_t)sz);
    if (!buf || fread(buf, 1, (size_t)sz, fp) != (size_t)sz) {
        fclose(fp);
        free(buf);
        return 0;
    }
    fclose(fp);
    hjo_image_t *im = (hjo_image_t *)malloc(sizeof(hjo_image_t));
    int ok = 0;
    if (im && hjo_parse(buf, (size_t)sz, im) == HJO_OK) {
        size_t npx = (size_t)im->width * im->height;
        uint8_t *rgb = (uint8_t *)calloc(npx, 3);
        uint8_t *out = (uint8_t *)malloc(hjo_bmp_size(im->width, im->height));
        if (rgb && out && hjo_decode(buf, (size_t)sz, rgb, NULL, NULL, NULL) == HJO_OK) {
            hjo_bmp_pack(rgb, im->width, im->height, out);
            FILE *fo = fopen(bmp, "wb");
            if (fo) {
                ok = fwrite(out, 1, hjo_bmp_size(im->width, im->height), fo) ==
                     hjo_bmp_size(im->width, im->height);
                fclose(fo);
            }
        }
        free(rgb);
        free(out);
    }
    free(im);
    free(buf);
    return ok ? 1 : 0;
}
