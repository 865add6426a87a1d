// xr_api.cpp -- error plumbing and device queries for the C ABI (host-only translation unit).
#include "xr_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void xr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* xr_last_error(void) { return g_err; }
extern "C" int xr_version(void) { return 100; }

extern "C" int xr_device_cus(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    xr_set_error("xr_device_cus: no HIP device");
    return XR_E_NODEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    xr_set_error("xr_device_cus: hipGetDeviceProperties failed");
    return XR_E_NODEVICE;
  }
  return prop.multiProcessorCount;
}
