// Definitions shared by the MFMA screen (newref_topk_screen.hip) and the exact refine
// (newref_refine.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256;      // 4 waves per workgroup
constexpr int TGT = 128;     // target rows per workgroup (32 per wave)
constexpr int CT = 64;       // candidate rows per main-loop iteration (2 MFMA tiles)
constexpr int CAP = 1024;    // shortlist capacity per target
constexpr int LIM = CAP - CT;

struct RowInfo {
  float nb;  // |a~|^2
  float e;   // >= |a - a~|
  float L;   // >= |a_lo|
  float N;   // >= |a~|
};

struct ScreenGlobals {
  unsigned long long amax_bits;  // max |x - c| over finite entries (double bits)
  unsigned int e_max, L_max, N_max;  // float bits, finite rows only
  unsigned int n_overflow;
  unsigned int uinv;             // 0xffffffff - min(norm float bits >> 20) over finite rows
};


struct ChrTab {
  int n_chr;
  int64_t cum[32];
};

struct wcx_ctx;
int wcx_refine_launch(wcx_ctx *ctx, const double *Xr, int S, int Sp, const ChrTab &tab,
                      int64_t row_begin, int64_t n_rows, const unsigned char *searched,
                      const uint2 *sl, const int *cnt_out, const unsigned int *flags,
                      const int *perm, int k, int32_t *d_out_idx, double *d_out_dist,
                      ScreenGlobals *glob);
