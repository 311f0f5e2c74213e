// gs_internal.h — helpers shared by the host translation units of libgs3d_hip.so
#pragma once
#include "../../include/gs3d.h"

// records the thread-local error details (gs_last_error) and returns `code`
gs_status gs_fail(gs_status code, uint64_t a, uint64_t b, uint64_t c, const char *fmt, ...)
    __attribute__((format(printf, 5, 6)));
