#include "sr_common.h"
static thread_local char g_err[512] = "";
void sr_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* sr_last_error(void) { return g_err; }
extern "C" int sr_version(void) { return 200; }
// hash of every source / header / flag this library was built from (build.py passes it); the Python loader compares it with the
// sources lying next to the library, so a stale binary is rebuilt or refused, never silently tested
#ifndef SR_SRC_HASH
#define SR_SRC_HASH "unhashed"
#endif
extern "C" const char* sr_source_hash(void) { return SR_SRC_HASH; }
extern "C" int sr_device_sync(void) {
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { sr_set_error("sync: %s", hipGetErrorString(e)); return SR_ERR_LAUNCH; }
  return SR_OK;
}
