// hjd_host.cpp — host-side glue on either side of the device path: the JFIF header walk that
// produces the kernels' inputs (N1) and the BMP/file plumbing that consumes their output (N2).
// No pixel or entropy arithmetic happens on the host.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/hjd.h"
#include "../csrc/hjd_device.h"
#include "hjd_internal.h"

static thread_local int t_last_status = HJD_OK;
void hjd_set_last_status(int s) { t_last_status = s; }

extern "C" int hjd_last_status(void) { return t_last_status; }

extern "C" const char *hjd_status_string(int s)
{
    switch (s) {
    case HJD_OK: return "ok";
    case HJD_ERR_NOT_JPEG: return "not a JPEG (no SOI / bogus marker)";
    case HJD_ERR_UNSUPPORTED: return "outside the reference's decodable domain";
    case HJD_ERR_CORRUPT: return "corrupt or truncated stream";
    case HJD_ERR_DIRTY_RESTART: return "restart stream outside the clean-resync domain";
    case HJD_ERR_NO_DEVICE: return "no usable HIP device (there is no CPU fallback)";
    case HJD_ERR_ARG: return "bad argument";
    case HJD_ERR_IO: return "file I/O failed";
    case HJD_ERR_INTERNAL: return "internal / HIP runtime error";
    default: return "unknown status";
    }
}

static inline unsigned be16(const uint8_t *p) { return ((unsigned)p[0] << 8) | p[1]; }

// One DHT payload (bits[1..16], vals[]) as GenHuffCodes / BuildHuffmanTable would take it (openjpg.cpp:48-98).
int hjd_validate_huff_spec(const hjd_huff_spec *h, int is_dc, int flags)
{
    if (!h->defined) return HJD_ERR_UNSUPPORTED;
    unsigned total = 0, code = 0;
    for (int k = 1; k <= 16; k++) {
        total += h->bits[k];
        code += h->bits[k];
        if (code > (1u << k)) return HJD_ERR_CORRUPT; // canonical codes must stay inside each length (GenHuffCodes :48-66)
        code <<= 1;
    }
    if (total > 255) return HJD_ERR_UNSUPPORTED; // uint8_t scan index never terminates (loadjpg.cpp:350-355)
    if (is_dc) {
        if (h->bits[16] && !(flags & HJD_PARSE_CONFORMANT)) return HJD_ERR_UNSUPPORTED; // DC search stops at 15 bits (loadjpg.cpp:562)
        for (unsigned k = 0; k < total; k++)
            if (h->vals[k] > 15) return HJD_ERR_UNSUPPORTED; // reader holds at most 2 new bytes (:468-482)
    }
    return HJD_OK;
}

int hjd_validate_jpeg_data(const hjd_jpeg_data *j, int flags)
{
    if (!j || !j->stream) return HJD_ERR_ARG;
    if (j->stream_len == 0) return HJD_ERR_CORRUPT;
    if (j->stream_len >= (1u << 28)) return HJD_ERR_UNSUPPORTED;
    if (!j->width || !j->height || j->width > 65535 || j->height > 65535) return HJD_ERR_UNSUPPORTED;
    const unsigned hf = j->hfactor, vf = j->vfactor;
    if (!(hf == 1 || hf == 2 || hf == 4) || !(vf == 1 || vf == 2 || vf == 4) || hf * vf > 4) return HJD_ERR_UNSUPPORTED;
    for (int c = 1; c <= 3; c++) {
        if (j->dc_table_index[c] > 3 || j->ac_table_index[c] > 3) return HJD_ERR_ARG;
        int st = hjd_validate_huff_spec(&j->htdc[j->dc_table_index[c]], 1, flags);
        if (st != HJD_OK) return st;
        st = hjd_validate_huff_spec(&j->htac[j->ac_table_index[c]], 0, flags);
        if (st != HJD_OK) return st;
    }
    return HJD_OK;
}

// Header walk with the observable behaviour of JpegParseHeader/ParseJFIF (openjpg.cpp:371-496) on
// the inputs the reference decodes correctly; everything it would mis-handle (SURVEY.md App. A.2,
// App. E) is rejected with a status instead.
extern "C" int hjd_parse_header(const uint8_t *buf, size_t n, hjd_jpeg_data *o) { return hjd_parse_header_ex(buf, n, o, 0); }

extern "C" int hjd_parse_header_ex(const uint8_t *buf, size_t n, hjd_jpeg_data *o, int flags)
{
    if (!buf || !o) return HJD_ERR_ARG;
    memset(o, 0, sizeof(*o));
    if (n < 4 || buf[0] != 0xFF || buf[1] != 0xD8) return HJD_ERR_NOT_JPEG; // openjpg.cpp:481-486

    uint8_t dqt[4][64];
    bool dqt_def[4] = {false, false, false, false};
    bool have_sof = false;
    bool comp_def[4] = {false, false, false, false};
    unsigned comp_h[4] = {0, 0, 0, 0}, comp_v[4] = {0, 0, 0, 0};
    size_t pos = 2;

    while (true) { // openjpg.cpp:381-462
        if (pos >= n) return HJD_ERR_CORRUPT;
        if (buf[pos++] != 0xFF) return HJD_ERR_NOT_JPEG; // :383-386 "bogus jpeg format"
        while (pos < n && buf[pos] == 0xFF) pos++;        // :389-392 fill bytes
        if (pos + 3 > n) return HJD_ERR_CORRUPT;
        const unsigned marker = buf[pos++];
        const uint8_t *seg = buf + pos; // the 2-byte length field
        const size_t left = n - pos;
        unsigned seglen = be16(seg);

        switch (marker) {
        case 0xD8:
        case 0xD9: // :433-439 embedded SOI/EOI: no payload
            seglen = 0;
            break;
        case 0xDB: { // ParseDQT :120-155
            if (seglen < 2 || seglen > left) return HJD_ERR_CORRUPT;
            size_t q = 2;
            while (q < seglen) {
                if (seglen - q < 65) return HJD_ERR_CORRUPT;
                const unsigned pq_tq = seg[q];
                if (pq_tq >> 4) return HJD_ERR_UNSUPPORTED;       // 16-bit tables: reference mis-parses (:136-143)
                if ((pq_tq & 15) > 3) return HJD_ERR_UNSUPPORTED; // :145-148
                memcpy(dqt[pq_tq & 15], seg + q + 1, 64);         // kept in zig-zag order (:102-116)
                dqt_def[pq_tq & 15] = true;
                q += 65;
            }
            break;
        }
        case 0xC0: { // ParseSOF :310-367
            if (seglen < 8 || seglen > left) return HJD_ERR_CORRUPT;
            if (seg[2] != 8) return HJD_ERR_UNSUPPORTED;
            o->height = be16(seg + 3);
            o->width = be16(seg + 5);
            const unsigned nc = seg[7];
            if (nc != 3) return HJD_ERR_UNSUPPORTED; // :180-183 "We only support YCbCr"
            if (seglen < 8 + 3 * nc) return HJD_ERR_CORRUPT;
            for (unsigned i = 0; i < nc; i++) {
                const uint8_t *c = seg + 8 + 3 * i;
                const unsigned id = c[0], tq = c[2];
                if (id < 1 || id > 3) return HJD_ERR_UNSUPPORTED;       // components are indexed by id (:343)
                if (tq > 3 || !dqt_def[tq]) return HJD_ERR_UNSUPPORTED; // table is copied at SOF time (:347-350)
                comp_h[id] = c[1] >> 4; // :345
                comp_v[id] = c[1] & 15; // :344
                memcpy(o->qtable[id], dqt[tq], 64);
                comp_def[id] = true;
            }
            have_sof = true;
            break;
        }
        case 0xC4: { // ParseDHT :234-305
            if (seglen < 2 || seglen > left) return HJD_ERR_CORRUPT;
            size_t q = 2;
            while (q < seglen) {
                if (seglen - q < 17) return HJD_ERR_CORRUPT;
                const unsigned tc_th = seg[q];
                if ((tc_th & 15) > 3) return HJD_ERR_UNSUPPORTED; // :273-276
                hjd_huff_spec *h = (tc_th & 0xF0) ? &o->htac[tc_th & 15] : &o->htdc[tc_th & 15];
                unsigned total = 0;
                h->bits[0] = 0;
                for (int k = 1; k <= 16; k++) {
                    h->bits[k] = seg[q + k];
                    total += h->bits[k];
                }
                if (total > 255) return HJD_ERR_UNSUPPORTED; // uint8_t scan index never terminates (loadjpg.cpp:350-355)
                if (seglen - q < 17 + total) return HJD_ERR_CORRUPT;
                memset(h->vals, 0, sizeof(h->vals));
                memcpy(h->vals, seg + q + 17, total);
                h->defined = 1;
                const int vst = hjd_validate_huff_spec(h, !(tc_th & 0xF0), flags);
                if (vst != HJD_OK) {
                    h->defined = 0;
                    return vst;
                }
                q += 17 + total;
            }
            break;
        }
        case 0xDD: // :441-446 — the reference stores the LENGTH field and only tests it > 0
            if (seglen > left) return HJD_ERR_CORRUPT;
            o->restart_interval = (int)seglen;
            o->restart_mcus = seglen >= 4 ? be16(seg + 2) : 0;
            break;
        case 0xDA: { // ParseSOS :160-229
            if (!have_sof) return HJD_ERR_UNSUPPORTED; // SOF1/SOF2/...: "Unknown marker", then SIGFPE (loadjpg.cpp:1153)
            if (left < 3) return HJD_ERR_CORRUPT;
            const unsigned ns = seg[2];
            if (ns != 3) return HJD_ERR_UNSUPPORTED;
            if (left < 3 + 2 * ns + 3) return HJD_ERR_CORRUPT;
            for (unsigned i = 0; i < ns; i++) {
                const unsigned id = seg[3 + 2 * i], sel = seg[4 + 2 * i];
                if (id != i + 1) return HJD_ERR_UNSUPPORTED; // block order is fixed Y,Cb,Cr (loadjpg.cpp:945-997)
                if ((sel & 15) > 3 || (sel >> 4) > 3) return HJD_ERR_UNSUPPORTED;
                o->ac_table_index[id] = sel & 15; // :212
                o->dc_table_index[id] = sel >> 4; // :213
                if (!o->htac[sel & 15].defined || !o->htdc[sel >> 4].defined) return HJD_ERR_UNSUPPORTED;
            }
            const size_t first = pos + 3 + 2 * ns + 3; // :186-190 and the "+3" of :222
            o->stream = buf + first;
            o->stream_len = n - first;
            o->header_bytes = first;
            goto walked;
        }
        default: // APPn, COM, unknown: skipped by length (:448-458)
            if (seglen > left) return HJD_ERR_CORRUPT;
            break;
        }
        pos += seglen; // :461
    }
walked:
    if (!comp_def[1] || !comp_def[2] || !comp_def[3]) return HJD_ERR_UNSUPPORTED;
    // one Cb + one Cr block per MCU (loadjpg.cpp:974-996); m_Y holds four luma blocks (loadjpg.h:165);
    // the chroma index math of loadjpg.cpp:912 is exact only for factors 1, 2, 4
    if (comp_h[2] != 1 || comp_v[2] != 1 || comp_h[3] != 1 || comp_v[3] != 1) return HJD_ERR_UNSUPPORTED;
    const unsigned hf = comp_h[1], vf = comp_v[1];
    if (!(hf == 1 || hf == 2 || hf == 4) || !(vf == 1 || vf == 2 || vf == 4) || hf * vf > 4) return HJD_ERR_UNSUPPORTED;
    if (!o->width || !o->height) return HJD_ERR_UNSUPPORTED;
    if (o->stream_len == 0) return HJD_ERR_CORRUPT;
    o->hfactor = (uint8_t)hf;
    o->vfactor = (uint8_t)vf;
    return HJD_OK;
}

// ---- Huffman table construction (GenHuffCodes / BuildHuffmanTable, openjpg.cpp:48-98) ----------

// Fills primary table `ti` of the set from one DHT payload; long codes go to secondary tables.
// Returns false when the set runs out of secondary tables.
static bool build_table(HjdTableSet &ts, int ti, const hjd_huff_spec &h, bool is_dc)
{
    memset(ts.lut[ti], 0, sizeof(ts.lut[ti]));
    memset(ts.lut_s[ti], 0, sizeof(ts.lut_s[ti]));
    // The two forms are built alike, each on its own index width (HJD_LUT_BITS / HJD_LUT_BITS_S: the prefixes that need a
    // second-level table differ with the width): put(form, index, entry), sec_new(form) -> index of a fresh second-level table or -1.
    auto fill = [&](const unsigned bits, const bool sync_form) -> bool {
        const unsigned sec_bits = 16 - bits;
        unsigned code = 0, idx = 0;
        for (int k = 1; k <= 16; k++) {
            for (unsigned j = 0; j < h.bits[k]; j++, idx++, code++) {
                const unsigned sym = h.vals[idx];
                unsigned s, adv;
                if (is_dc) { // DC symbol = number of extra bits (loadjpg.cpp:616)
                    s = sym & 15;
                    adv = 1;
                } else { // AC run/size (loadjpg.cpp:768-808)
                    s = sym & 15;
                    const unsigned r = sym >> 4;
                    if (s) adv = r + 1;
                    else adv = r == 0 ? 64 : (r == 15 ? 16 : 0);
                }
                const uint16_t e = sync_form ? (uint16_t)((k + s) | (adv << 5) | ((!is_dc && s) ? 1u << 14 : 0u)) // synchronisation form
                                             : (uint16_t)((k + s) | (s << 5) | (adv << 9)); // (k + s <= 31: the walk admits 16-bit codes, 15 value bits)
                if ((unsigned)k <= bits) {
                    const unsigned lo = code << (bits - k), cnt = 1u << (bits - k);
                    for (unsigned q = 0; q < cnt; q++) {
                        if (sync_form) ts.lut_s[ti][lo + q][0] = ts.lut_s[ti][lo + q][1] = e;
                        else ts.lut[ti][lo + q] = e;
                    }
                } else {
                    const unsigned prefix = code >> (k - bits);
                    uint16_t pe = sync_form ? ts.lut_s[ti][prefix][0] : ts.lut[ti][prefix];
                    if (pe == 0) {
                        uint32_t &nsec = sync_form ? ts.nsec_s : ts.nsec;
                        if (nsec >= HJD_MAX_SEC) return false;
                        if (sync_form) memset(ts.sec_s[nsec], 0, sizeof(ts.sec_s[0]));
                        else memset(ts.sec[nsec], 0, sizeof(ts.sec[0]));
                        pe = (uint16_t)((nsec + 1) << 5);
                        if (sync_form) ts.lut_s[ti][prefix][0] = ts.lut_s[ti][prefix][1] = pe;
                        else ts.lut[ti][prefix] = pe;
                        nsec++;
                    }
                    const unsigned si = (pe >> 5) - 1;
                    const unsigned rem = code & ((1u << (k - bits)) - 1);
                    const unsigned sbits = sec_bits - (k - bits);
                    for (unsigned q = 0; q < (1u << sbits); q++) {
                        if (sync_form) ts.sec_s[si][(rem << sbits) + q] = e;
                        else ts.sec[si][(rem << sbits) + q] = e;
                    }
                }
            }
            code <<= 1;
        }
        return true;
    };
    if (!fill(HJD_LUT_BITS, false) || !fill(HJD_LUT_BITS_S, true)) return false;
    if (!is_dc) {
        // two AC symbols in one step of the synchronisation pass (HjdTableSet::lut_s): the bits after the first symbol
        // (code + value bits) that the window still holds are looked up again, left-aligned and zero-padded; the pair
        // exists if that finds a code no longer than the bits that are really there
        for (unsigned i = 0; i < HJD_LUT_SIZE_S; i++) {
            const unsigned e1 = ts.lut_s[ti][i][0];
            const unsigned tot1 = e1 & 31u, adv1 = (e1 >> 5) & 127u;
            if (tot1 == 0 || tot1 >= HJD_LUT_BITS_S || adv1 == 0 || adv1 > 16) continue; // long code, no room, invalid symbol, EOB
            const unsigned rem = HJD_LUT_BITS_S - tot1;
            const unsigned i2 = (i << tot1) & (HJD_LUT_SIZE_S - 1u);
            const unsigned w2 = ts.lut_s[ti][i2][0]; // (the second symbol's one-symbol entry: total bits | advance << 5 | records << 14)
            const unsigned t2 = w2 & 31u, adv2 = (w2 >> 5) & 127u;
            if (t2 == 0 || adv2 == 0) continue;
            // its code must lie wholly inside the bits that are really there: code length = total - value bits, from the final-pass
            // form of the same symbol where the widths agree; here recovered from the canonical code itself
            unsigned l2 = 0;
            {   // length of the code that the rem bits of i2's top start with
                unsigned code = 0, first = 0;
                const unsigned top = i2 << (16 - HJD_LUT_BITS_S); // 16-bit left-aligned view
                for (int k = 1; k <= 16 && !l2; k++) {
                    code = top >> (16 - k);
                    if (code >= first && code < first + h.bits[k]) l2 = (unsigned)k;
                    first = (first + h.bits[k]) << 1;
                }
            }
            if (l2 == 0 || l2 > rem) continue;
            const unsigned tot = tot1 + t2, rec = (e1 >> 14) + (w2 >> 14);
            if (tot > 31u) continue;
            ts.lut_s[ti][i][1] = (uint16_t)(tot | ((adv1 + adv2) << 5) | (rec << 14));
        }
    }
    return true;
}

// One table set per distinct (DC,AC) x (Y,Cb,Cr) configuration; identical payloads share a table.
bool hjd_build_tabset(HjdTableSet &ts, const hjd_jpeg_data &j)
{
    memset(&ts, 0, sizeof(ts));
    const hjd_huff_spec *specs[HJD_SLOTS];
    for (int c = 1; c <= 3; c++) {
        specs[(c - 1) * 2 + 0] = &j.htdc[j.dc_table_index[c]];
        specs[(c - 1) * 2 + 1] = &j.htac[j.ac_table_index[c]];
    }
    for (int sl = 0; sl < HJD_SLOTS; sl++) {
        int ti = -1;
        for (int q = 0; q < sl; q++)
            if ((q & 1) == (sl & 1) && !memcmp(specs[q]->bits, specs[sl]->bits, 17) && !memcmp(specs[q]->vals, specs[sl]->vals, 256)) {
                ti = (int)((ts.slot_pack >> (q * 4)) & 7);
                break;
            }
        if (ti < 0) {
            ti = (int)ts.nlut++;
            if (!build_table(ts, ti, *specs[sl], (sl & 1) == 0)) return false;
        }
        ts.slot_pack |= (uint32_t)ti << (sl * 4);
    }
    return true;
}


// ---- sizes / BMP (WriteBMP24, openjpg.cpp:504-570) ------------------------------------------------

extern "C" size_t hjd_rgb_bytes(unsigned w, unsigned h) { return (size_t)w * h * 3; }

extern "C" size_t hjd_bmp_bytes(unsigned w, unsigned h)
{
    const unsigned pad = (4 - (w * 3) % 4) % 4; // :534-535
    return (size_t)w * h * 3 + (size_t)h * pad + 54; // :541
}

extern "C" void hjd_free(void *p) { free(p); }

bool hjd_read_file(const char *path, std::vector<uint8_t> &out)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return false;
    fseek(fp, 0, SEEK_END);
    long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (sz < 0) {
        fclose(fp);
        return false;
    }
    out.resize((size_t)sz);
    bool ok = sz == 0 || fread(out.data(), 1, (size_t)sz, fp) == (size_t)sz;
    fclose(fp);
    return ok;
}

bool hjd_write_file(const char *path, const uint8_t *data, size_t n)
{
    FILE *fp = fopen(path, "wb");
    if (!fp) return false;
    bool ok = fwrite(data, 1, n, fp) == n;
    ok = (fclose(fp) == 0) && ok;
    return ok;
}
