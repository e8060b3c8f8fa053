// Instantiations of the symmetric screen kernel (see screen_sym.h): K = 896, 1024 (765 .. 1020 samples).
#include "screen_sym.h"

int wcx_sym_launch_k5(int nk, int ctg, int lb, int ring, const SymArgs &a, unsigned grid, size_t lds,
                      hipStream_t st) {
  WCX_SYM_TRY(56, 1, 1, 2)
  WCX_SYM_TRY(64, 1, 1, 2)
  return -1;
}
