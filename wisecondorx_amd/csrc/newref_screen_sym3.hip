// Instantiations of the symmetric screen kernel (see screen_sym.h): K = 320 .. 512.
#include "screen_sym.h"

int wcx_sym_launch_k3(int nk, int ctg, int lb, int ring, const SymArgs &a, unsigned grid, size_t lds,
                      hipStream_t st) {
  WCX_SYM_TRY(20, 1, 2, 2)
  WCX_SYM_TRY(24, 1, 2, 2)
  WCX_SYM_TRY(28, 1, 2, 2)
  WCX_SYM_TRY(32, 1, 2, 2)
  return -1;
}
