// Library identity + device check.
#include <string.h>

#include "common.hpp"

extern "C" {

const char* d3f_version(void) { return "d3feat-hip 0.1 (gfx950)"; }

int d3f_device_arch_ok(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

}  // extern "C"
