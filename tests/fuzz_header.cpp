// fuzz_header.cpp — header-walk fuzz for the sanitizer build (`make sanitize`, CPU only; not collected by pytest).
// Everything the host does with bytes it has not checked yet: hjd_parse_header_ex (the JFIF walk that replaces ParseJFIF,
// openjpg.cpp:371-474), hjd_validate_jpeg_data on what it returns AND on structs mangled after parsing (the caller-filled path of
// hjd_jpeg_decode_hw), and the table builder (hjd_build_tabset; GenHuffCodes / BuildHuffmanTable, openjpg.cpp:48-98) on every struct
// that passes validation.  Deterministic (xorshift, seed on the command line); built with -fsanitize=address,undefined: any
// out-of-bounds read, overflow or misaligned access aborts the run.
//   usage: fuzz_header <iterations> <seed> <file.jpg>...      prints a summary line; exit 0
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../hls-jpeg-decoder_amd/csrc/hjd_device.h"
#include "../hls-jpeg-decoder_amd/host/hjd_internal.h"

static uint64_t g_s;
static inline uint64_t rnd()
{
    g_s ^= g_s << 13;
    g_s ^= g_s >> 7;
    g_s ^= g_s << 17;
    return g_s;
}

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const long iters = atol(argv[1]);
    g_s = (uint64_t)atoll(argv[2]) * 0x9E3779B97F4A7C15ull + 1u;
    std::vector<std::vector<uint8_t>> seeds;
    for (int i = 3; i < argc; i++) {
        std::vector<uint8_t> f;
        if (!hjd_read_file(argv[i], f) || f.size() < 4) {
            fprintf(stderr, "cannot read %s\n", argv[i]);
            return 2;
        }
        seeds.push_back(f);
    }
    HjdTableSet *ts = new HjdTableSet;
    long parsed = 0, valid = 0, built = 0, mangled_valid = 0;
    for (long it = 0; it < iters; it++) {
        const std::vector<uint8_t> &src = seeds[rnd() % seeds.size()];
        // header = everything up to the scan (plus a little of it); exact-size heap copy so that any over-read is caught
        size_t hdr = src.size();
        for (size_t q = 2; q + 1 < src.size(); q++)
            if (src[q] == 0xFF && src[q + 1] == 0xDA) {
                hdr = q + 2 + 12 + 64 < src.size() ? q + 2 + 12 + 64 : src.size();
                break;
            }
        size_t n = hdr;
        const unsigned mode = (unsigned)(rnd() % 8);
        if (mode == 0) n = 1 + rnd() % hdr; // truncated anywhere
        uint8_t *buf = (uint8_t *)malloc(n);
        memcpy(buf, src.data(), n);
        const unsigned flips = mode == 1 ? 0u : 1u + (unsigned)(rnd() % 6);
        for (unsigned f = 0; f < flips; f++) {
            const size_t pos = rnd() % n;
            switch (rnd() % 4) {
            case 0: buf[pos] ^= (uint8_t)(1u << (rnd() % 8)); break;
            case 1: buf[pos] = (uint8_t)rnd(); break;
            case 2: buf[pos] = 0xFF; break;
            default: buf[pos] = (uint8_t)(rnd() % 3 ? 0 : 0xC4); break;
            }
        }
        hjd_jpeg_data j;
        const int flags = (rnd() & 1) ? HJD_PARSE_CONFORMANT : 0;
        if (hjd_parse_header_ex(buf, n, &j, flags) == HJD_OK) {
            parsed++;
            if (j.stream < buf || j.stream + j.stream_len > buf + n) abort(); // the scan must lie inside the buffer
            if (hjd_validate_jpeg_data(&j, flags) == HJD_OK) {
                valid++;
                if (hjd_build_tabset(*ts, j)) built++;
            }
            // the caller-filled path: a parsed struct with fields overwritten at random goes through validation only
            hjd_jpeg_data m = j;
            uint8_t *raw = reinterpret_cast<uint8_t *>(&m);
            const size_t tab0 = offsetof(hjd_jpeg_data, dc_table_index), tab1 = offsetof(hjd_jpeg_data, stream);
            for (unsigned f = 0; f < 1u + rnd() % 8; f++) raw[tab0 + rnd() % (tab1 - tab0)] = (uint8_t)rnd();
            if (rnd() & 1) m.hfactor = (uint8_t)rnd(), m.vfactor = (uint8_t)rnd();
            if (hjd_validate_jpeg_data(&m, flags) == HJD_OK) {
                mangled_valid++;
                (void)hjd_build_tabset(*ts, m);
            }
        }
        free(buf);
    }
    printf("fuzz_header: %ld inputs, %ld parsed, %ld valid, %ld table sets built, %ld mangled structs passed validation; clean\n", iters, parsed,
           valid, built, mangled_valid);
    delete ts;
    return 0;
}
