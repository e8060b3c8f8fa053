// Instantiations of the MFMA screen kernel (see screen_kernel.h), one group of K sizes per unit.
#include "screen_kernel.h"

int wcx_screen_launch_k1(const ScreenCfg &c, const ScreenArgs &a, unsigned grid, size_t lds,
                         hipStream_t st) {
  WCX_SCREEN_TRY(1, 2, 1, 4, 3, 3, false)
  WCX_SCREEN_TRY(2, 2, 1, 4, 3, 3, false)
  WCX_SCREEN_TRY(3, 2, 1, 4, 3, 3, false)
  WCX_SCREEN_TRY(4, 2, 1, 4, 3, 3, false)
  return -1;
}
