// Library identity + device check.
#include <string.h>

#include "common.hpp"

extern "C" {

const char* d3f_version(void) { return "d3feat-hip 0.1 (gfx950)"; }

// returns 1 for gfx950, 0 for another architecture, negative HIP error code (negated) if the query itself failed
int d3f_device_arch_ok(void) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return -(int)e;
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return -(int)e;
  return strstr(prop.gcnArchName, "gfx950") != nullptr ? 1 : 0;
}

int d3f_device_arch_name(char* out, int n) {
  int dev = 0;
  if (!out || n < 1) return D3F_EINVAL;
  out[0] = 0;
  if (hipGetDevice(&dev) != hipSuccess) return D3F_ELAUNCH;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return D3F_ELAUNCH;
  strncpy(out, prop.gcnArchName, (size_t)n - 1);
  out[n - 1] = 0;
  return D3F_OK;
}

}  // extern "C"
