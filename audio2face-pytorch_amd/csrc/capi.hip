// C-ABI plumbing: version and thread-local error text.
#include "a2f_common.h"

namespace a2f {
char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
int set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace a2f

extern "C" int a2f_version(void) { return A2F_VERSION; }
extern "C" const char* a2f_last_error(void) { return a2f::err_buf(); }
