// Instantiations of the hub-count estimator (see screen_count.h): K = 448 .. 640.
#include "screen_count.h"

int wcx_count_launch_k2(int nk, int ctg, int lb, int ring, const CountArgs &a, unsigned grid, size_t lds,
                        hipStream_t st) {
  WCX_COUNT_TRY(28, 1, 2, 2)
  WCX_COUNT_TRY(32, 1, 2, 2)
  WCX_COUNT_TRY(40, 1, 1, 2)
  return -1;
}
