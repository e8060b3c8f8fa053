// Definitions shared by the MFMA screen (newref_topk_screen.hip) and the exact refine
// (newref_refine.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256;      // threads of the helper kernels (prep, merge, refine)
constexpr int CT = 64;       // sweep positions are padded to a multiple of this (gmask granule)
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


// One workgroup of the screen: consecutive target rows of one chromosome.
struct ScreenBlock {
  int64_t row0;
  int32_t nrows;  // <= 32 TT WPB of the launched configuration
  int32_t chr;    // chromosome index of the target rows
  int64_t cs, ce;
};

struct ScreenArgs {
  const half8 *F;
  const RowInfo *info;
  const ScreenGlobals *glob;
  const int *perm, *rowpos;
  const unsigned int *gmask;
  const ScreenBlock *blocks;
  uint2 *sl;
  int *cnt;
  unsigned int *flags;
  float *g_state;
  unsigned long long *stats;
  int64_t row_begin, n_rows_all;
  int64_t g_start;          // this launch visits the candidate groups [g_start, g_start + g_count)
  int g_count;
  int k;                    // refsize
  int cut_k, cut_mode;      // in-sweep cuts: rank and mode (sampled pre-pass: r, 1; main pass: k, 0)
  int trig;                 // shortlist length that triggers an in-sweep cut
  int end_cut;              // cut of every target at the end of the launch: 0 none, 1 = end of the
                            // sampled pre-pass (estimate from rank cut_k), 2 = final (exact k-th)
  int first, dbg, n_seg, n_blocks;
};

// Screen kernel configuration: K = 16 nk, ctg candidate sub-tiles per iteration, tt target tiles
// per wave, wpb waves per workgroup (targets per workgroup = 32 tt wpb), lb = waves per SIMD the
// registers are budgeted for, ring = slots of the LDS-DMA staging ring (0 = register staging into a
// double buffer); prof = phase accounting.
struct ScreenCfg { int nk, ctg, tt, wpb, lb, ring, prof; };
// Launchers of the instantiated configurations (newref_screen_k*.hip); return -1 if `c` is not
// instantiated there, else a hipError_t.
int wcx_screen_launch_k1(const ScreenCfg &c, const ScreenArgs &a, unsigned grid, size_t lds, hipStream_t st);
int wcx_screen_launch_k2(const ScreenCfg &c, const ScreenArgs &a, unsigned grid, size_t lds, hipStream_t st);
int wcx_screen_launch_k3(const ScreenCfg &c, const ScreenArgs &a, unsigned grid, size_t lds, hipStream_t st);
int wcx_screen_launch_k4(const ScreenCfg &c, const ScreenArgs &a, unsigned grid, size_t lds, hipStream_t st);
int wcx_screen_launch_k5(const ScreenCfg &c, const ScreenArgs &a, unsigned grid, size_t lds, hipStream_t st);
int wcx_screen_launch_k6(const ScreenCfg &c, const ScreenArgs &a, unsigned grid, size_t lds, hipStream_t st);

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
