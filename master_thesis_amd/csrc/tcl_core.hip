// Error reporting and version of libtcl_hip.so.
#include "tcl_common.h"

namespace tcl {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(TCL_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return TCL_OK;
}

}  // namespace tcl

extern "C" int tcl_version(void) { return TCL_VERSION; }
extern "C" const char *tcl_last_error(void) { return tcl::g_err; }
