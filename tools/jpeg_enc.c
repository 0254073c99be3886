/*
 * jpeg_enc.c — small baseline-JPEG ENCODER + synthetic image generator.
 *
 * Test/bench tooling only (the product decodes; nothing in the reference encodes).
 * It exists because the GPU box has no network and encoder output bytes are
 * version-dependent: bench.py and the full-size tests need 4:2:0 / 4:2:2 / 4:4:4
 * baseline JPEGs (SOF0, 8-bit, component ids 1/2/3, DQT before SOF0, standard
 * Annex-K Huffman tables, optional DRI) of arbitrary size, generated in place.
 *
 * Build: gcc -O2 -fPIC -shared -o libhjd_enc.so jpeg_enc.c -lm
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* zig-zag position -> natural index */
static const uint8_t UNZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static const uint8_t QL[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                               14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                               18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                               49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t QC[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                               24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                               99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                               99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

static const uint8_t DC_L_BITS[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t DC_C_BITS[17] = {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t DC_VALS[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t AC_L_BITS[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t AC_L_VALS[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t AC_C_BITS[17] = {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t AC_C_VALS[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

typedef struct {
    uint16_t code[256];
    uint8_t len[256];
} enc_tab_t;

static void make_tab(enc_tab_t *t, const uint8_t *bits, const uint8_t *vals)
{
    memset(t, 0, sizeof(*t));
    unsigned code = 0;
    int k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < bits[l]; i++) {
            t->code[vals[k]] = (uint16_t)code++;
            t->len[vals[k]] = (uint8_t)l;
            k++;
        }
        code <<= 1;
    }
}

typedef struct {
    uint8_t *out;
    size_t cap, pos;
    uint32_t acc;
    int nacc;
    int overflow;
} bw_t;

static void put_byte(bw_t *w, uint8_t b)
{
    if (w->pos < w->cap) w->out[w->pos] = b;
    else w->overflow = 1;
    w->pos++;
}
static void put_bits(bw_t *w, unsigned v, int n)
{
    if (n == 0) return;
    w->acc = (w->acc << n) | (v & ((1u << n) - 1u));
    w->nacc += n;
    while (w->nacc >= 8) {
        uint8_t b = (uint8_t)(w->acc >> (w->nacc - 8));
        put_byte(w, b);
        if (b == 0xFF) put_byte(w, 0);
        w->nacc -= 8;
    }
}
static void flush_bits(bw_t *w)
{
    if (w->nacc > 0) put_bits(w, (1u << (8 - w->nacc)) - 1u, 8 - w->nacc); /* pad with 1s */
    w->acc = 0;
    w->nacc = 0;
}
static void put_word(bw_t *w, unsigned v)
{
    put_byte(w, (uint8_t)(v >> 8));
    put_byte(w, (uint8_t)v);
}

static int nbits_of(int v)
{
    if (v < 0) v = -v;
    int n = 0;
    while (v) {
        n++;
        v >>= 1;
    }
    return n;
}

static float CT[8][8]; /* CT[u][x] = C(u)/2 * cos((2x+1)u*pi/16) */
static int ct_init = 0;
static void init_ct(void)
{
    if (ct_init) return;
    for (int u = 0; u < 8; u++)
        for (int x = 0; x < 8; x++)
            CT[u][x] = (float)((u == 0 ? sqrt(0.5) : 1.0) * 0.5 * cos((2 * x + 1) * u * M_PI / 16.0));
    ct_init = 1;
}

static void fdct_quant(const float *px /*64, level-shifted*/, const uint8_t *q, int16_t *zz)
{
    float tmp[64], F[64];
    for (int y = 0; y < 8; y++)
        for (int u = 0; u < 8; u++) {
            float s = 0;
            for (int x = 0; x < 8; x++) s += px[y * 8 + x] * CT[u][x];
            tmp[y * 8 + u] = s;
        }
    for (int v = 0; v < 8; v++)
        for (int u = 0; u < 8; u++) {
            float s = 0;
            for (int y = 0; y < 8; y++) s += tmp[y * 8 + u] * CT[v][y];
            F[v * 8 + u] = s;
        }
    for (int k = 0; k < 64; k++) {
        int nat = UNZZ[k];
        zz[k] = (int16_t)lrintf(F[nat] / (float)q[nat]);
    }
}

/* ---- optimised Huffman tables (ITU T.81 Annex K.2, written from the text) ---------------- */

/* freq[0..255] symbol counts -> bits[1..16], vals[] of a length-limited canonical code.  A reserved
 * 257th symbol keeps the all-ones code unused.  maxlen = 16 (AC) or 15 (DC: the reference's DC code
 * search stops at 15 bits, loadjpg.cpp:562). */
static int gen_optimal_table(const long *freq_in, int maxlen, uint8_t *bits /*17*/, uint8_t *vals /*256*/)
{
    long freq[257];
    int codesize[257], others[257];
    uint8_t cnt[64];
    memset(cnt, 0, sizeof(cnt));
    for (int i = 0; i < 256; i++) freq[i] = freq_in[i];
    freq[256] = 1;
    for (int i = 0; i < 257; i++) {
        codesize[i] = 0;
        others[i] = -1;
    }
    for (;;) {
        int c1 = -1, c2 = -1;
        long v = 1000000000L;
        for (int i = 0; i <= 256; i++)
            if (freq[i] && freq[i] <= v) {
                v = freq[i];
                c1 = i;
            }
        v = 1000000000L;
        for (int i = 0; i <= 256; i++)
            if (freq[i] && freq[i] <= v && i != c1) {
                v = freq[i];
                c2 = i;
            }
        if (c2 < 0) break;
        freq[c1] += freq[c2];
        freq[c2] = 0;
        codesize[c1]++;
        while (others[c1] >= 0) {
            c1 = others[c1];
            codesize[c1]++;
        }
        others[c1] = c2;
        codesize[c2]++;
        while (others[c2] >= 0) {
            c2 = others[c2];
            codesize[c2]++;
        }
    }
    for (int i = 0; i <= 256; i++)
        if (codesize[i]) {
            if (codesize[i] > 63) return 0;
            cnt[codesize[i]]++;
        }
    for (int i = 63; i > maxlen; i--) /* Figure K.3: move pairs of too-long codes up */
        while (cnt[i] > 0) {
            int j = i - 2;
            while (j > 0 && cnt[j] == 0) j--;
            if (j <= 0) return 0;
            cnt[i] -= 2;
            cnt[i - 1]++;
            cnt[j + 1] += 2;
            cnt[j]--;
        }
    {
        int i = maxlen;
        while (i > 0 && cnt[i] == 0) i--;
        if (i <= 0) return 0;
        cnt[i]--; /* the reserved code point */
    }
    memset(bits, 0, 17);
    for (int i = 1; i <= 16; i++) bits[i] = cnt[i];
    int k = 0;
    for (int l = 1; l <= 63; l++)
        for (int i = 0; i < 256; i++)
            if (codesize[i] == l) vals[k++] = (uint8_t)i;
    return k;
}

/* same code lengths, symbols handed out in a different order: still a valid prefix code for the same
 * symbol set, and a different DHT payload (used to give Cb and Cr tables of their own) */
static void permute_vals(const uint8_t *vals, int n, uint8_t *out, int salt)
{
    memcpy(out, vals, (size_t)n);
    for (int i = 0; i + 1 < n; i += 2 + (salt & 1)) {
        uint8_t t = out[i];
        out[i] = out[i + 1];
        out[i + 1] = t;
    }
}

typedef struct hje_params {
    unsigned w, h, hf, vf;
    int quality;            /* libjpeg scaling of the Annex K tables, when qtab == NULL */
    const uint8_t *qtab;    /* optional 3 x 64 quantiser tables, natural order: Y, Cb, Cr */
    int three_dqt;          /* 1: three DQT segments (ids 0,1,2; Cb -> 1, Cr -> 2); 0: two, Cb and Cr share id 1 */
    int huff_mode;          /* 0: the four Annex K tables; 1: six tables, Cb/Cr get permuted Annex K tables;
                               2: six tables optimised per component; 3: four optimised tables (Y / chroma) */
    unsigned restart_mcus;
    const uint8_t *rgb;     /* h*w*3, when coefs == NULL */
    const int16_t *coefs;   /* optional: blocks in scan order, 64 zig-zag values each, [0] = the DC VALUE the
                               decoder must end up with (differences are taken here, int16 wrap) */
} hje_params;

typedef struct {
    uint8_t bits[17];
    uint8_t vals[256];
    int n;
    enc_tab_t tab;
} huff_def_t;

static void def_from(huff_def_t *d, const uint8_t *bits, const uint8_t *vals)
{
    memcpy(d->bits, bits, 17);
    d->n = 0;
    for (int l = 1; l <= 16; l++) d->n += bits[l];
    memset(d->vals, 0, 256);
    memcpy(d->vals, vals, (size_t)d->n);
    make_tab(&d->tab, d->bits, d->vals);
}

static void count_block(const int16_t *zz, int diff, long *fdc, long *fac)
{
    fdc[nbits_of(diff)]++;
    int run = 0;
    for (int k = 1; k < 64; k++) {
        int v = zz[k];
        if (v == 0) {
            run++;
            continue;
        }
        while (run > 15) {
            fac[0xF0]++;
            run -= 16;
        }
        fac[(run << 4) | nbits_of(v)]++;
        run = 0;
    }
    if (run > 0) fac[0]++;
}

/* returns 0 when a symbol has no code in the table in use */
static int encode_block_diff(bw_t *w, const int16_t *zz, int diff, const enc_tab_t *dc, const enc_tab_t *ac)
{
    int n = nbits_of(diff);
    if (n > 15 || !dc->len[n]) return 0;
    put_bits(w, dc->code[n], dc->len[n]);
    if (n) put_bits(w, (unsigned)(diff < 0 ? diff - 1 : diff), n);
    int run = 0;
    for (int k = 1; k < 64; k++) {
        int v = zz[k];
        if (v == 0) {
            run++;
            continue;
        }
        while (run > 15) {
            if (!ac->len[0xF0]) return 0;
            put_bits(w, ac->code[0xF0], ac->len[0xF0]);
            run -= 16;
        }
        int s = nbits_of(v);
        int rs = (run << 4) | s;
        if (s > 15 || !ac->len[rs]) return 0;
        put_bits(w, ac->code[rs], ac->len[rs]);
        put_bits(w, (unsigned)(v < 0 ? v - 1 : v), s);
        run = 0;
    }
    if (run > 0) {
        if (!ac->len[0]) return 0;
        put_bits(w, ac->code[0], ac->len[0]);
    }
    return 1;
}

/*
 * General form.  Returns the number of bytes the stream needs (<= cap when it fitted), 0 on bad
 * arguments or when a coefficient cannot be coded with the chosen tables.
 */
size_t hje_encode_ex(const hje_params *P, uint8_t *out, size_t cap)
{
    const unsigned w = P->w, h = P->h, hf = P->hf, vf = P->vf;
    if (!w || !h || w > 65535 || h > 65535 || !hf || !vf || hf * vf > 4) return 0;
    if (!P->rgb && !P->coefs) return 0;
    init_ct();
    int quality = P->quality;
    if (quality < 1) quality = 1;
    if (quality > 100) quality = 100;
    int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    uint8_t qt[3][64]; /* natural order */
    for (int i = 0; i < 64; i++) {
        if (P->qtab) {
            for (int c = 0; c < 3; c++) qt[c][i] = P->qtab[c * 64 + i] ? P->qtab[c * 64 + i] : 1;
        } else {
            int a = (QL[i] * scale + 50) / 100, b = (QC[i] * scale + 50) / 100;
            qt[0][i] = (uint8_t)(a < 1 ? 1 : (a > 255 ? 255 : a));
            qt[1][i] = (uint8_t)(b < 1 ? 1 : (b > 255 ? 255 : b));
            /* a third table that differs from Cb's everywhere it can: coarser, and transposed */
            int t = i / 8 + (i % 8) * 8;
            int c = (QC[t] * scale * 3 / 2 + 50) / 100 + 1;
            qt[2][i] = P->three_dqt ? (uint8_t)(c < 1 ? 1 : (c > 255 ? 255 : c)) : qt[1][i];
        }
    }
    const unsigned mw = 8 * hf, mh = 8 * vf;
    const unsigned mcux = (w + mw - 1) / mw, mcuy = (h + mh - 1) / mh;
    const unsigned nby = hf * vf, nb = nby + 2;
    const size_t nblocks = (size_t)mcux * mcuy * nb;

    /* pass 1: all quantised blocks in scan order */
    int16_t *cf = NULL;
    const int16_t *C = P->coefs;
    if (!C) {
        cf = (int16_t *)malloc(nblocks * 64 * sizeof(int16_t));
        float *Y = (float *)malloc(sizeof(float) * mw * mh);
        float *Cb = (float *)malloc(sizeof(float) * mw * mh);
        float *Cr = (float *)malloc(sizeof(float) * mw * mh);
        if (!cf || !Y || !Cb || !Cr) {
            free(cf); free(Y); free(Cb); free(Cr);
            return 0;
        }
        float blk[64];
        size_t bi = 0;
        for (unsigned my = 0; my < mcuy; my++)
            for (unsigned mx = 0; mx < mcux; mx++) {
                for (unsigned y = 0; y < mh; y++) {
                    unsigned sy = my * mh + y;
                    if (sy >= h) sy = h - 1;
                    for (unsigned x = 0; x < mw; x++) {
                        unsigned sx = mx * mw + x;
                        if (sx >= w) sx = w - 1;
                        const uint8_t *p = P->rgb + ((size_t)sy * w + sx) * 3;
                        float r = p[0], g = p[1], b = p[2];
                        Y[y * mw + x] = 0.299f * r + 0.587f * g + 0.114f * b - 128.0f;
                        Cb[y * mw + x] = -0.168736f * r - 0.331264f * g + 0.5f * b;
                        Cr[y * mw + x] = 0.5f * r - 0.418688f * g - 0.081312f * b;
                    }
                }
                for (unsigned by = 0; by < vf; by++)
                    for (unsigned bx = 0; bx < hf; bx++) {
                        for (int y = 0; y < 8; y++)
                            for (int x = 0; x < 8; x++) blk[y * 8 + x] = Y[(by * 8 + y) * mw + bx * 8 + x];
                        fdct_quant(blk, qt[0], cf + 64 * bi++);
                    }
                for (int c = 0; c < 2; c++) {
                    const float *src = c ? Cr : Cb;
                    for (int y = 0; y < 8; y++)
                        for (int x = 0; x < 8; x++) {
                            float s = 0;
                            for (unsigned dy = 0; dy < vf; dy++)
                                for (unsigned dx = 0; dx < hf; dx++) s += src[(y * vf + dy) * mw + x * hf + dx];
                            blk[y * 8 + x] = s / (float)(hf * vf);
                        }
                    fdct_quant(blk, qt[1 + c], cf + 64 * bi++);
                }
            }
        free(Y); free(Cb); free(Cr);
        C = cf;
    }

    /* tables: slot = component*2 + isAC */
    huff_def_t T[6];
    int tid[6]; /* DHT identifier (Th) of each slot */
    if (P->huff_mode == 2 || P->huff_mode == 3) {
        long f[6][256];
        memset(f, 0, sizeof(f));
        int pred[3] = {0, 0, 0};
        unsigned mc = 0;
        for (size_t b = 0; b < nblocks; b++) {
            const unsigned m = (unsigned)(b % nb);
            const int c = m < nby ? 0 : (int)(m - nby + 1);
            if (P->restart_mcus && m == 0) {
                if (mc && mc % P->restart_mcus == 0) pred[0] = pred[1] = pred[2] = 0;
                mc++;
            }
            const int16_t *zz = C + 64 * b;
            const int diff = (int16_t)(zz[0] - pred[c]);
            pred[c] = zz[0];
            const int g = P->huff_mode == 3 && c == 2 ? 1 : c; /* mode 3: Cb and Cr pooled */
            count_block(zz, diff, f[g * 2], f[g * 2 + 1]);
        }
        for (int sl = 0; sl < 6; sl++) {
            const int src = P->huff_mode == 3 && sl >= 4 ? sl - 2 : sl;
            uint8_t bits[17], vals[256];
            if (src != sl) {
                T[sl] = T[src];
                continue;
            }
            memset(vals, 0, sizeof(vals));
            if (!gen_optimal_table(f[sl], (sl & 1) ? 16 : 15, bits, vals)) {
                free(cf);
                return 0;
            }
            def_from(&T[sl], bits, vals);
        }
    } else {
        def_from(&T[0], DC_L_BITS, DC_VALS);
        def_from(&T[1], AC_L_BITS, AC_L_VALS);
        def_from(&T[2], DC_C_BITS, DC_VALS);
        def_from(&T[3], AC_C_BITS, AC_C_VALS);
        if (P->huff_mode == 1) {
            uint8_t pv[256];
            permute_vals(DC_VALS, 12, pv, 0);
            def_from(&T[4], DC_C_BITS, pv);
            permute_vals(AC_C_VALS, 162, pv, 1);
            def_from(&T[5], AC_C_BITS, pv);
        } else {
            T[4] = T[2];
            T[5] = T[3];
        }
    }
    const int six = P->huff_mode == 1 || P->huff_mode == 2;
    for (int sl = 0; sl < 6; sl++) tid[sl] = six ? sl / 2 : (sl < 2 ? 0 : 1);

    bw_t bw = {out, cap, 0, 0, 0, 0};
    bw_t *W = &bw;
    put_word(W, 0xFFD8);
    /* APP0 JFIF */
    put_word(W, 0xFFE0);
    put_word(W, 16);
    put_byte(W, 'J'); put_byte(W, 'F'); put_byte(W, 'I'); put_byte(W, 'F'); put_byte(W, 0);
    put_word(W, 0x0101);
    put_byte(W, 0);
    put_word(W, 1);
    put_word(W, 1);
    put_byte(W, 0); put_byte(W, 0);
    /* DQT x2 or x3 (zig-zag order) */
    for (int t = 0; t < (P->three_dqt ? 3 : 2); t++) {
        put_word(W, 0xFFDB);
        put_word(W, 67);
        put_byte(W, (uint8_t)t);
        for (int k = 0; k < 64; k++) put_byte(W, qt[t][UNZZ[k]]);
    }
    /* SOF0 */
    put_word(W, 0xFFC0);
    put_word(W, 17);
    put_byte(W, 8);
    put_word(W, h);
    put_word(W, w);
    put_byte(W, 3);
    put_byte(W, 1); put_byte(W, (uint8_t)((hf << 4) | vf)); put_byte(W, 0);
    put_byte(W, 2); put_byte(W, 0x11); put_byte(W, 1);
    put_byte(W, 3); put_byte(W, 0x11); put_byte(W, (uint8_t)(P->three_dqt ? 2 : 1));
    /* DHT x4 or x6 */
    for (int sl = 0; sl < (six ? 6 : 4); sl++) {
        put_word(W, 0xFFC4);
        put_word(W, (unsigned)(19 + T[sl].n));
        put_byte(W, (uint8_t)(((sl & 1) << 4) | tid[sl]));
        for (int l = 1; l <= 16; l++) put_byte(W, T[sl].bits[l]);
        for (int i = 0; i < T[sl].n; i++) put_byte(W, T[sl].vals[i]);
    }
    if (P->restart_mcus) {
        put_word(W, 0xFFDD);
        put_word(W, 4);
        put_word(W, P->restart_mcus & 0xFFFF);
    }
    /* SOS */
    put_word(W, 0xFFDA);
    put_word(W, 12);
    put_byte(W, 3);
    for (int c = 0; c < 3; c++) {
        put_byte(W, (uint8_t)(c + 1));
        put_byte(W, (uint8_t)((tid[c * 2] << 4) | tid[c * 2 + 1]));
    }
    put_byte(W, 0); put_byte(W, 63); put_byte(W, 0);

    /* pass 2: entropy coding */
    int pred[3] = {0, 0, 0};
    unsigned mcu_count = 0, rst = 0;
    int ok = 1;
    for (size_t b = 0; b < nblocks && ok; b++) {
        const unsigned m = (unsigned)(b % nb);
        const int c = m < nby ? 0 : (int)(m - nby + 1);
        if (m == 0) {
            if (P->restart_mcus && mcu_count && (mcu_count % P->restart_mcus) == 0) {
                flush_bits(W);
                put_byte(W, 0xFF);
                put_byte(W, (uint8_t)(0xD0 + (rst & 7)));
                rst++;
                pred[0] = pred[1] = pred[2] = 0;
            }
            mcu_count++;
        }
        const int16_t *zz = C + 64 * b;
        const int diff = P->coefs ? (int)(int16_t)(zz[0] - pred[c]) : zz[0] - pred[c];
        pred[c] = zz[0];
        ok = encode_block_diff(W, zz, diff, &T[c * 2].tab, &T[c * 2 + 1].tab);
    }
    flush_bits(W);
    put_word(W, 0xFFD9);
    free(cf);
    return ok ? bw.pos : 0;
}

/*
 * rgb: h*w*3 bytes.  quality 1..100 (libjpeg scaling).  hf,vf: luma sampling (chroma 1x1).
 * restart_mcus: 0 = no DRI; otherwise DRI interval in MCUs with RSTn markers.  Two DQT, the four
 * Annex K Huffman tables.  Returns the number of bytes the stream needs (<= cap when it fitted), 0 on bad args.
 */
size_t hje_encode(const uint8_t *rgb, unsigned w, unsigned h, int quality, unsigned hf, unsigned vf,
                  unsigned restart_mcus, uint8_t *out, size_t cap)
{
    hje_params P;
    memset(&P, 0, sizeof(P));
    P.w = w; P.h = h; P.hf = hf; P.vf = vf;
    P.quality = quality;
    P.restart_mcus = restart_mcus;
    P.rgb = rgb;
    return hje_encode_ex(&P, out, cap);
}

/* ---- deterministic synthetic content ------------------------ */

static uint64_t rng_state;
static inline uint64_t rng_next(void)
{ /* splitmix64 */
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double rng_uniform(void) { return (double)(rng_next() >> 11) * (1.0 / 9007199254740992.0); }

/* per channel: 128 + 60 sin(a0 x+p0) cos(a1 y+p1) + 40 sin(a2 (x+y)+p2) + N(0, sigma^2) */
void hje_synth_rgb(uint8_t *rgb, unsigned w, unsigned h, uint64_t seed, double sigma)
{
    rng_state = seed * 0x9E3779B97F4A7C15ull + 12345;
    for (int c = 0; c < 3; c++) {
        double a[3], p[3];
        for (int i = 0; i < 3; i++) {
            a[i] = 0.005 + 0.045 * rng_uniform();
            p[i] = 6.283185307179586 * rng_uniform();
        }
        for (unsigned y = 0; y < h; y++) {
            double cy = cos(a[1] * y + p[1]);
            for (unsigned x = 0; x < w; x += 2) {
                double u1 = rng_uniform(), u2 = rng_uniform();
                if (u1 < 1e-300) u1 = 1e-300;
                double rad = sqrt(-2.0 * log(u1)) * sigma;
                double n0 = rad * cos(6.283185307179586 * u2), n1 = rad * sin(6.283185307179586 * u2);
                for (unsigned k = 0; k < 2 && x + k < w; k++) {
                    unsigned xx = x + k;
                    double v = 128.0 + 60.0 * sin(a[0] * xx + p[0]) * cy + 40.0 * sin(a[2] * (xx + y) + p[2]) +
                               (k ? n1 : n0);
                    int iv = (int)lrint(v);
                    rgb[((size_t)y * w + xx) * 3 + c] = (uint8_t)(iv < 0 ? 0 : (iv > 255 ? 255 : iv));
                }
            }
        }
    }
}
