// Error plumbing + version of the C ABI (include/lecone.h).
#include <cstdarg>
#include <cstdio>
#include <hip/hip_runtime_api.h>
#include "../../include/lecone.h"

namespace lec {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s", what, hipGetErrorString(e));
  return LEC_E_HIP;
}
}  // namespace lec

extern "C" const char* lec_last_error(void) { return lec::g_err; }
extern "C" int lec_abi_version(void) { return 28; }
