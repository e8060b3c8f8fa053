// Instantiations of the MFMA screen kernel (see screen_kernel.h), one group of K sizes per unit.
#include "screen_kernel.h"

int wcx_screen_launch_k3(const ScreenCfg &c, const ScreenArgs &a, unsigned grid, size_t lds,
                         hipStream_t st) {
  WCX_SCREEN_TRY(10, 2, 1, 4, 2, 2, false)
  WCX_SCREEN_TRY(12, 2, 1, 4, 2, 2, false)
  WCX_SCREEN_TRY(14, 2, 1, 4, 2, 2, false)
  WCX_SCREEN_TRY(16, 2, 1, 4, 2, 2, false)
  return -1;
}
