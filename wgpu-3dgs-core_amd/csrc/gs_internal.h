// gs_internal.h — helpers shared by the host translation units of libgs3d_hip.so
#pragma once
#include "../../include/gs3d.h"

// records the thread-local error details (gs_last_error) and returns `code`
gs_status gs_fail(gs_status code, uint64_t a, uint64_t b, uint64_t c, const char *fmt, ...)
    __attribute__((format(printf, 5, 6)));

#ifdef __cplusplus
#include <vector>
// gs_spz.cpp, for the device load path in gs3d.hip: validates the header of a DECOMPRESSED SPZ payload
// and its length, returns the byte offsets of the six columns (positions, alphas, colors, scales,
// rotations, sh) and the SH coefficients per channel; errors as gs_spz_decode_decompressed
gs_status gs_spz_payload_layout(const void *bytes, size_t len, gs_spz_header *header_out, size_t col_offset[6],
                                uint32_t *ncoef_out);
// the bounded gunzip of gs_spz_decode
gs_status gs_spz_gunzip(const void *bytes, size_t len, std::vector<uint8_t> &out);
#endif
