// mc_internal.hpp -- the few hooks the library's own translation units share beside the public C ABI (include/mc_hip.h).
// Hidden visibility: none of this is exported (tests/test_abi.py checks that `nm -D` shows mc_* of the header only).
#pragma once
#include "../../include/mc_hip.h"

// sets the calling thread's mc_last_error() text and returns `code`
int mc_internal_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// the result of the context's last sweep as the mc_copy_* calls see it (null before the first sweep)
const mc_result* mc_internal_last(const mc_context* ctx);
// the HIP device the context lives on
int mc_internal_device(const mc_context* ctx);
