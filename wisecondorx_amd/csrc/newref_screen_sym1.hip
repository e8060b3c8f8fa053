// Instantiations of the symmetric screen kernel (see screen_sym.h): K = 16 .. 128.
#include "screen_sym.h"

int wcx_sym_launch_k1(int nk, int ctg, int lb, int ring, const SymArgs &a, unsigned grid, size_t lds,
                      hipStream_t st) {
  WCX_SYM_TRY(1, 2, 3, 3)
  WCX_SYM_TRY(2, 2, 3, 3)
  WCX_SYM_TRY(3, 2, 3, 3)
  WCX_SYM_TRY(4, 2, 3, 3)
  WCX_SYM_TRY(5, 2, 3, 3)
  WCX_SYM_TRY(6, 2, 3, 3)
  WCX_SYM_TRY(7, 2, 3, 3)
  WCX_SYM_TRY(8, 2, 3, 3)
  return -1;
}
