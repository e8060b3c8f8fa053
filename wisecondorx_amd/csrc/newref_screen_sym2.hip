// Instantiations of the symmetric screen kernel (see screen_sym.h): K = 160 .. 256.
#include "screen_sym.h"

int wcx_sym_launch_k2(int nk, int ctg, int lb, int ring, const SymArgs &a, unsigned grid, size_t lds,
                      hipStream_t st) {
  WCX_SYM_TRY(10, 2, 2, 2)
  WCX_SYM_TRY(12, 2, 2, 2)
  WCX_SYM_TRY(14, 2, 2, 2)
  WCX_SYM_TRY(16, 2, 2, 2)
  return -1;
}
