// shim_check.cpp — test harness (build container only): the reference's OWN header parser, then
// include/hjd_reference_shim.h, against hjd_parse_header.  Compiled by tests/test_reference_shim.py with the
// reference's sources on the include path (openjpg.cpp is included textually: JpegParseHeader is `inline` and
// TU-local); the reference's JpegDecodeHW is provided here as the shim would provide it.
//   usage: shim_check <file.jpg> [<Ri>]   -> prints "OK" or the first difference; exit code 0 / 1
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "openjpg.cpp" // the reference's parser, where it lies

#include "hjd_reference_shim.h"

static unsigned g_dri;

int JpegDecodeHW(stJpegData *jdata, unsigned int h, unsigned int w, unsigned char hF, unsigned char vF)
{
    return hjd_shim_jpeg_decode_hw_ri(jdata, g_dri, h, w, hF, vF);
}

#define DIFF(what)                              \
    do {                                        \
        printf("DIFF %s\n", what);              \
        return 1;                               \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *fp = fopen(argv[1], "rb");
    if (!fp) return 2;
    std::vector<unsigned char> buf(BUF_SIZE + 65536, 0);
    const size_t n = fread(buf.data(), 1, BUF_SIZE, fp);
    fclose(fp);
    g_dri = argc > 2 ? (unsigned)atoi(argv[2]) : 0;

    stJpegData *jd = new stJpegData();
    stImageInfo *ji = new stImageInfo();
    fclose(stdout); // the reference's parser prints
    stdout = fopen("/dev/null", "w");
    if (JpegParseHeader(jd, ji, buf.data(), (unsigned)n) < 0) return 3;
    stdout = stderr;

    hjd_jpeg_data a, b;
    hjd_shim_from_reference(jd, ji, g_dri, &a);
    const int st = hjd_parse_header(buf.data(), n, &b);
    if (st != HJD_OK) {
        printf("hjd_parse_header refused the file: %d\n", st);
        return 1;
    }
    if (a.width != b.width || a.height != b.height) DIFF("dimensions");
    if (a.hfactor != b.hfactor || a.vfactor != b.vfactor) DIFF("sampling factors");
    for (int c = 1; c <= 3; c++) {
        if (memcmp(a.qtable[c], b.qtable[c], 64)) DIFF("quantiser table");
        if (a.dc_table_index[c] != b.dc_table_index[c] || a.ac_table_index[c] != b.ac_table_index[c]) DIFF("table selectors");
        const hjd_huff_spec *pa[2] = {&a.htdc[a.dc_table_index[c]], &a.htac[a.ac_table_index[c]]};
        const hjd_huff_spec *pb[2] = {&b.htdc[b.dc_table_index[c]], &b.htac[b.ac_table_index[c]]};
        for (int k = 0; k < 2; k++) {
            if (!pa[k]->defined || !pb[k]->defined) DIFF("table not defined");
            if (memcmp(pa[k]->bits + 1, pb[k]->bits + 1, 16)) DIFF("code length counts");
            int total = 0;
            for (int l = 1; l <= 16; l++) total += pb[k]->bits[l];
            if (memcmp(pa[k]->vals, pb[k]->vals, (size_t)total)) DIFF("symbol values");
        }
    }
    if ((a.restart_interval > 0) != (b.restart_interval > 0)) DIFF("restart flag");
    if (a.stream_len > b.stream_len) DIFF("scan length runs past the file");
    if (a.stream_len < 2 || a.stream[a.stream_len - 2] != 0xFF || a.stream[a.stream_len - 1] != 0xD9) DIFF("scan does not end at EOI");
    if (memcmp(a.stream, b.stream, a.stream_len)) DIFF("scan bytes");
    hjd_layout_info li;
    if (hjd_layout_query(&a, 0, 0, HJD_OUT_RGB, &li) != HJD_OK) DIFF("shim output fails the library's validation");
    printf("OK %ux%u %zu scan bytes (file holds %zu after the header)\n", a.width, a.height, a.stream_len, b.stream_len);
    return 0;
}
