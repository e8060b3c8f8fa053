// Instantiations of the MFMA screen kernel (see screen_kernel.h): K = 896, 1024 (765 .. 1020 samples).
#include "screen_kernel.h"

int wcx_screen_launch_k8(const ScreenCfg &c, const ScreenArgs &a, unsigned grid, size_t lds,
                         hipStream_t st) {
  WCX_SCREEN_TRY(56, 1, 1, 4, 1, 2, false)
  WCX_SCREEN_TRY(64, 1, 1, 4, 1, 2, false)
  return -1;
}
