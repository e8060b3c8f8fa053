// Instantiations of the MFMA screen kernel (see screen_kernel.h), one group of K sizes per unit.
#include "screen_kernel.h"

int wcx_screen_launch_k5(const ScreenCfg &c, const ScreenArgs &a, unsigned grid, size_t lds,
                         hipStream_t st) {
  WCX_SCREEN_TRY(7, 2, 1, 4, 3, 3, false)
  WCX_SCREEN_TRY(7, 2, 1, 4, 3, 3, true)
  WCX_SCREEN_TRY(7, 2, 1, 4, 3, 0, false)
  WCX_SCREEN_TRY(7, 2, 2, 4, 2, 3, false)
  return -1;
}
