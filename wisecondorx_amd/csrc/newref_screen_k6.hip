// Instantiations of the MFMA screen kernel (see screen_kernel.h), one group of K sizes per unit.
#include "screen_kernel.h"

int wcx_screen_launch_k6(const ScreenCfg &c, const ScreenArgs &a, unsigned grid, size_t lds,
                         hipStream_t st) {
  WCX_SCREEN_TRY(32, 1, 1, 4, 2, 2, false)
  WCX_SCREEN_TRY(32, 1, 1, 4, 2, 2, true)
  WCX_SCREEN_TRY(32, 2, 1, 8, 2, 2, false)
  WCX_SCREEN_TRY(32, 2, 1, 8, 2, 2, true)
  WCX_SCREEN_TRY(32, 1, 1, 4, 2, 0, false)
  return -1;
}
