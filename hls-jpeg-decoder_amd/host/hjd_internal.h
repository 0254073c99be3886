// hjd_internal.h — shared between the host translation units (not part of the C ABI).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

void hjd_set_last_status(int s);
bool hjd_read_file(const char *path, std::vector<uint8_t> &out);
bool hjd_write_file(const char *path, const uint8_t *data, size_t n);
