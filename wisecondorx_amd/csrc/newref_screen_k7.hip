// Instantiations of the MFMA screen kernel (see screen_kernel.h): K = 640, 768 (509 .. 764 samples),
// one wave per SIMD (the target fragments take 160 / 192 registers of the unified VGPR + AGPR file).
#include "screen_kernel.h"

int wcx_screen_launch_k7(const ScreenCfg &c, const ScreenArgs &a, unsigned grid, size_t lds,
                         hipStream_t st) {
  WCX_SCREEN_TRY(40, 1, 1, 4, 1, 2, false)
  WCX_SCREEN_TRY(48, 1, 1, 4, 1, 2, false)
  return -1;
}
