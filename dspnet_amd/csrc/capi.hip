// Library-wide C ABI helpers (error string, version).
#include "dspn_common.h"
#include "../../include/dspn_multibox.h"

namespace dspn {
char *last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace dspn

extern "C" {
const char *dspn_last_error(void) { return dspn::last_error_buf(); }
int dspn_abi_version(void) { return 1; }
}
