// Instantiations of the MFMA screen kernel (see screen_kernel.h), one group of K sizes per unit.
#include "screen_kernel.h"

int wcx_screen_launch_k2(const ScreenCfg &c, const ScreenArgs &a, unsigned grid, size_t lds,
                         hipStream_t st) {
  WCX_SCREEN_TRY(5, 2, 1, 4, 3, 3, false)
  WCX_SCREEN_TRY(6, 2, 1, 4, 3, 3, false)
  WCX_SCREEN_TRY(8, 2, 1, 4, 3, 3, false)
  return -1;
}
