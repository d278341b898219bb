#include "dppm_common.h"

namespace dppm {
char* error_buffer() {
  static thread_local char buf[512] = "";
  return buf;
}
}  // namespace dppm

extern "C" const char* dppm_last_error(void) { return dppm::error_buffer(); }
extern "C" int dppm_abi_version(void) { return 1; }
