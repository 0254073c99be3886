// hjd_internal.h — shared between the host translation units (not part of the C ABI).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "../../include/hjd.h"

void hjd_set_last_status(int s);
// Everything batch_build / the table builder rely on, for a struct filled by hjd_parse_header OR by a
// caller of hjd_jpeg_decode_hw: table selectors in range and defined, DHT payloads that describe a
// prefix code (sum of counts <= 255, no length over-subscribed), DC symbols <= 15 and no 16-bit DC
// code (loadjpg.cpp:562), sampling factors the MCU loop supports, a non-empty scan.
// flags: HJD_PARSE_CONFORMANT admits what only the reference cannot take (16-bit DC codes)
int hjd_validate_huff_spec(const hjd_huff_spec *h, int is_dc, int flags);
int hjd_validate_jpeg_data(const hjd_jpeg_data *j, int flags);
// Decode tables of one distinct (DC, AC) x (Y, Cb, Cr) configuration (csrc/hjd_device.h: HjdTableSet) from validated DHT
// payloads; false when the set runs out of second-level tables.
struct HjdTableSet;
bool hjd_build_tabset(HjdTableSet &ts, const hjd_jpeg_data &j);
bool hjd_read_file(const char *path, std::vector<uint8_t> &out);
bool hjd_write_file(const char *path, const uint8_t *data, size_t n);
