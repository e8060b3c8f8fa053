// Instantiations of the hub-count estimator of the one-directional sweep (see screen_hub1.h): K = 320 .. 512.
#include "screen_hub1.h"
int wcx_hub1_launch_k3(int nk, int ctg, int lb, int ring, int trials, const Hub1Args &a, unsigned grid, size_t lds,
                       hipStream_t st) {
  WCX_HUB1_TRY(20, 1, 2, 2, 8)
  WCX_HUB1_TRY(24, 1, 2, 2, 8)
  WCX_HUB1_TRY(28, 1, 2, 2, 8)
  WCX_HUB1_TRY(32, 1, 2, 2, 8)
  return -1;
}
