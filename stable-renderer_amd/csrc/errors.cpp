#include "sr_common.h"
static thread_local char g_err[512] = "";
void sr_set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* sr_last_error(void) { return g_err; }
extern "C" int sr_version(void) { return 100; }
extern "C" int sr_device_sync(void) {
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { sr_set_error("sync: %s", hipGetErrorString(e)); return SR_ERR_LAUNCH; }
  return SR_OK;
}
