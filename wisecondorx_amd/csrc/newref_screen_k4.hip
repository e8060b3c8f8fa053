// Instantiations of the MFMA screen kernel (see screen_kernel.h), one group of K sizes per unit.
#include "screen_kernel.h"

int wcx_screen_launch_k4(const ScreenCfg &c, const ScreenArgs &a, unsigned grid, size_t lds,
                         hipStream_t st) {
  WCX_SCREEN_TRY(20, 1, 1, 4, 2, 2, false)
  WCX_SCREEN_TRY(24, 1, 1, 4, 2, 2, false)
  WCX_SCREEN_TRY(28, 1, 1, 4, 2, 2, false)
  return -1;
}
