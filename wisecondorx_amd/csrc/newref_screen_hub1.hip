// Instantiations of the hub-count estimator of the one-directional sweep (see screen_hub1.h): K = 80 .. 128.
#include "screen_hub1.h"
int wcx_hub1_launch_k1(int nk, int ctg, int lb, int ring, int trials, const Hub1Args &a, unsigned grid, size_t lds,
                       hipStream_t st) {
  WCX_HUB1_TRY(5, 2, 3, 3, 4)
  WCX_HUB1_TRY(5, 2, 3, 3, 8)
  WCX_HUB1_TRY(6, 2, 3, 3, 4)
  WCX_HUB1_TRY(6, 2, 3, 3, 8)
  WCX_HUB1_TRY(7, 2, 3, 3, 4)
  WCX_HUB1_TRY(7, 2, 3, 3, 8)
  WCX_HUB1_TRY(8, 2, 3, 3, 4)
  WCX_HUB1_TRY(8, 2, 3, 3, 8)
  return -1;
}
