// Instantiations of the hub-count estimator (see screen_count.h): K = 256 .. 384.
#include "screen_count.h"

int wcx_count_launch_k1(int nk, int ctg, int lb, int ring, const CountArgs &a, unsigned grid, size_t lds,
                        hipStream_t st) {
  WCX_COUNT_TRY(16, 2, 2, 2)
  WCX_COUNT_TRY(20, 1, 2, 2)
  WCX_COUNT_TRY(24, 1, 2, 2)
  return -1;
}
