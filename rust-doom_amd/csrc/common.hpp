// Shared by the host library and the HIP translation units: error reporting across the C ABI.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <string>

#include "rdoom.h"

namespace rdoom {
std::string &last_error_ref();
inline rdoom_status fail(rdoom_status code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error_ref() = buf;
  return code;
}
}  // namespace rdoom
