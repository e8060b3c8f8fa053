// Instantiations of the hub-count estimator of the one-directional sweep (see screen_hub1.h): K = 160 .. 256.
#include "screen_hub1.h"
int wcx_hub1_launch_k2(int nk, int ctg, int lb, int ring, int trials, const Hub1Args &a, unsigned grid, size_t lds,
                       hipStream_t st) {
  WCX_HUB1_TRY(10, 2, 2, 2, 4)
  WCX_HUB1_TRY(10, 2, 2, 2, 8)
  WCX_HUB1_TRY(12, 2, 2, 2, 4)
  WCX_HUB1_TRY(12, 2, 2, 2, 8)
  WCX_HUB1_TRY(14, 2, 2, 2, 4)
  WCX_HUB1_TRY(14, 2, 2, 2, 8)
  WCX_HUB1_TRY(16, 2, 2, 2, 8)
  WCX_HUB1_TRY(16, 2, 2, 2, 4)
  return -1;
}
