// Instantiations of the hub-count estimator (see screen_count.h): K = 80 .. 224 (the small-K symmetric
// sweep, WCX_SCREEN_SYM=2 / where it pays).
#include "screen_count.h"

int wcx_count_launch_k4(int nk, int ctg, int lb, int ring, const CountArgs &a, unsigned grid, size_t lds,
                        hipStream_t st) {
  WCX_COUNT_TRY(5, 2, 3, 3)
  WCX_COUNT_TRY(6, 2, 3, 3)
  WCX_COUNT_TRY(7, 2, 3, 3)
  WCX_COUNT_TRY(8, 2, 3, 3)
  WCX_COUNT_TRY(10, 2, 2, 2)
  WCX_COUNT_TRY(12, 2, 2, 2)
  WCX_COUNT_TRY(14, 2, 2, 2)
  return -1;
}
