// helpers shared by the translation units of the C ABI (sda_capi.cpp owns them)
#pragma once
#include <stdint.h>

#include "kernels.hpp"

int capi_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));   // sets sda_last_error(), returns code
int capi_make_mod(int64_t modulus, sda::ModParams& mod);                                  // validated Barrett / Lemire constants
int capi_device_ready();                                                                  // a device exists; make the selected one current
