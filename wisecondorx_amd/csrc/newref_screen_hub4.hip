// Instantiations of the hub-count estimator of the one-directional sweep (see screen_hub1.h): K = 640 .. 1024
// (509 .. 1020 samples: one wave per SIMD, target fragments on the unified VGPR + AGPR file).
#include "screen_hub1.h"
int wcx_hub1_launch_k4(int nk, int ctg, int lb, int ring, int trials, const Hub1Args &a, unsigned grid, size_t lds,
                       hipStream_t st) {
  WCX_HUB1_TRY(40, 1, 1, 2, 8)
  WCX_HUB1_TRY(48, 1, 1, 2, 8)
  WCX_HUB1_TRY(56, 1, 1, 2, 8)
  WCX_HUB1_TRY(64, 1, 1, 2, 8)
  return -1;
}
