// Version / architecture probes of libsivae_hip.
#include "common.h"
extern "C" int sivae_abi_version() { return SIVAE_ABI_VERSION; }
extern "C" const char* sivae_arch() { return "gfx950"; }
// number of visible HIP devices (0 on a box without a GPU); never throws
extern "C" int sivae_device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
