// Instantiations of the hub-count estimator (see screen_count.h): K = 768 .. 1024.
#include "screen_count.h"

int wcx_count_launch_k3(int nk, int ctg, int lb, int ring, const CountArgs &a, unsigned grid, size_t lds,
                        hipStream_t st) {
  WCX_COUNT_TRY(48, 1, 1, 2)
  WCX_COUNT_TRY(56, 1, 1, 2)
  WCX_COUNT_TRY(64, 1, 1, 2)
  return -1;
}
