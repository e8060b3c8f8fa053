// Instantiations of the symmetric screen kernel (see screen_sym.h): K = 640, 768 (509 .. 764 samples).
// More than 32 k-steps of target fragments do not fit beside a second wave: one wave per SIMD, the
// fragments spread over the unified VGPR + AGPR file.
#include "screen_sym.h"

int wcx_sym_launch_k4(int nk, int ctg, int lb, int ring, const SymArgs &a, unsigned grid, size_t lds,
                      hipStream_t st) {
  WCX_SYM_TRY(40, 1, 1, 2)
  WCX_SYM_TRY(48, 1, 1, 2)
  return -1;
}
