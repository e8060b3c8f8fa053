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
constexpr int CAP2 = 2048;   // list capacity per row in the symmetric sweep (fixed thresholds, no cuts);
constexpr int CAP2_BIG = 4096;   // ... for refsize > 448 (SymArgs::cap2 carries the one in use)
constexpr int KMAX_SCREEN = 1024;    // largest refsize of the MFMA paths (symmetric sweep)
constexpr int KMAX_ONE_DIR = 512;    // ... of the one-directional sweep: k + its filter margin (14 % at 15 kb, more
                                     // than 50 % on small noisy problems) must fit shortlists of LIM = 960 entries
constexpr int SMAX_SCREEN = 1020;    // most samples: K = 16 NK >= S + 4, NK <= 64 (target fragments in registers)

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
  unsigned int n_tiles;          // symmetric sweep: tiles in use (multiple of 4)
  unsigned int hub_key;          // rows with (float bits of |a|^2) >> 16 <= hub_key form the hub region
  unsigned int n_hub_tiles;      // tiles of the hub region (the head of the symmetric sweep order)
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
  const half8 *Ft;          // fragments the TARGET rows are read from (rowpos positions); = F unless the
                            // candidates are a separate sample array (symmetric path, pre-pass)
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
  int raw_est = 0;          // sampled pre-pass: the estimate is the r-th sample value itself (no filter margin)
  const unsigned int *gate = nullptr;   // not null: the launch does nothing unless *gate != 0 (second-attempt kernels)
};

// One chunk of streamed tiles of the symmetric sweep: work items = (quads q_first .. q_first + n_q - 1)
// x n_split shares of the chunk; n_split == 1: exclusive items (see k_screen_sym), `index` = the
// chunk's number = the position of the item among the exclusive items of its quad.
struct SymDesc {
  int c0, c1, q_first, n_q, n_split, item_base, index, pad;
};

// Arguments of the symmetric sweep (screen_sym.h).
struct SymArgs {
  const half8 *F;             // fragments, tile-major: half8[tile][NK][64]
  const unsigned int *tinfo;  // [tile][64]: 0..31 float bits of theta = -D/2 (+inf: nothing passes),
                              //             32..63 row id (-1 = padding)
  const float *tmin;          // [tile] min theta over the tile's rows
  const unsigned char *tchr;  // [tile] chromosome of the (pure) tile; 255 = no rows
  const ScreenGlobals *glob;  // n_tiles
  uint2 *sl;                  // [row][cap2]
  int cap2;                   // list capacity per row: CAP2 or CAP2_BIG
  int *cnt;                   // [row] entries appended (device-scope atomic)
  unsigned int *flags;        // [row] 1 = redo exactly
  unsigned long long *stats;
  uint4 *pool;                // records (row, partner position, d~ bits, -) for k_sym_regroup
  unsigned int *pool_head, *pool_ovf;
  unsigned int pool_cap;
  const SymDesc *desc;        // chunk descriptors, in queue order
  int n_desc, total_items;
  unsigned int *queue_head;   // next work item
  int *seq;                   // [quad] exclusive items of the quad that have finished
  int glist_cap;              // ints reserved for the visit list in LDS
  int dbg;
  int hub_appended = 0;       // the hub pass has appended the hits on hub candidates: skip that direction
  int force_records = 0;      // row-sharded sweep: every hit becomes a record (its row's list lives elsewhere)
  int part = 0, n_parts = 1;  // ... and this launch takes the work items  item % n_parts == part
  const unsigned int *gate = nullptr;   // not null: the launch does nothing unless *gate != 0
};
// Arguments of the hub-count estimator (screen_count.h).
struct CountArgs {
  const half8 *F;
  const unsigned char *tchr;
  ScreenGlobals *glob;        // n_tiles, n_hub_tiles
  const int *perm;            // sweep position -> row (-1 = padding)
  uint2 *sl;                  // second pass: the hub hits go to the rows' lists [row][cap2]
  int cap2;
  int append_pass = 0;        // 0 = the counting pass (thresholds), 1 = the second pass (needs sl)
  unsigned int *tinfo;        // out: [tile][64] theta | row id
  float *tmin;                // out: [tile]
  float *Dest;                // out: [row] threshold in screen-distance space
  int *cnt;                   // out: [row] = 0
  unsigned int *flags;        // out: rows without an estimate
  unsigned long long *stats;
  int need;                   // hub candidates wanted below the estimate
  int n1;                     // hub tiles of the moment phase
  int glist_cap;              // hub groups the visit list in LDS has room for
  const unsigned int *gate = nullptr;   // not null: the launch does nothing if *gate != 0
};
int wcx_count_launch_k1(int nk, int ctg, int lb, int ring, const CountArgs &a, unsigned grid, size_t lds, hipStream_t st);
int wcx_count_launch_k2(int nk, int ctg, int lb, int ring, const CountArgs &a, unsigned grid, size_t lds, hipStream_t st);
int wcx_count_launch_k3(int nk, int ctg, int lb, int ring, const CountArgs &a, unsigned grid, size_t lds, hipStream_t st);
int wcx_count_launch_k4(int nk, int ctg, int lb, int ring, const CountArgs &a, unsigned grid, size_t lds, hipStream_t st);
int wcx_sym_launch_k1(int nk, int ctg, int lb, int ring, const SymArgs &a, unsigned grid, size_t lds, hipStream_t st);
int wcx_sym_launch_k2(int nk, int ctg, int lb, int ring, const SymArgs &a, unsigned grid, size_t lds, hipStream_t st);
int wcx_sym_launch_k3(int nk, int ctg, int lb, int ring, const SymArgs &a, unsigned grid, size_t lds, hipStream_t st);
int wcx_sym_launch_k4(int nk, int ctg, int lb, int ring, const SymArgs &a, unsigned grid, size_t lds, hipStream_t st);
int wcx_sym_launch_k5(int nk, int ctg, int lb, int ring, const SymArgs &a, unsigned grid, size_t lds, hipStream_t st);

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
int wcx_screen_launch_k7(const ScreenCfg &c, const ScreenArgs &a, unsigned grid, size_t lds, hipStream_t st);
int wcx_screen_launch_k8(const ScreenCfg &c, const ScreenArgs &a, unsigned grid, size_t lds, hipStream_t st);

// LDS-DMA staging of one candidate group (NF fragment pieces of 1 KiB, laid out in LDS as in HBM) by the
// WPB waves of a workgroup: wave w copies the NFW = ceil(NF / WPB) CONSECUTIVE pieces from
// min(w NFW, NF - NFW) on (the last waves overlap their neighbours when NF is no multiple: same bytes,
// same place) -- one base address per four pieces, the others in the instruction's immediate offset,
// which the hardware adds on the global and on the LDS side alike.  No per-piece scalar address
// arithmetic or branches: ~3 scalar instructions per four pieces (the strided form -- piece w, w + WPB,
// ... -- cost 5-7 per piece, and a chain of branches where the piece kind depended on the wave).
// Every wave issues exactly NFW loads (the counted s_waitcnt of the rings relies on it).
template <int NF, int WPB>
__device__ __forceinline__ int stage_piece0(int wave_u) {
  constexpr int NFW = (NF + WPB - 1) / WPB;
  const int b = wave_u * NFW;
  return b < NF - NFW ? b : NF - NFW;
}
template <int NF, int WPB>
__device__ __forceinline__ void stage_group(const half8 *src_group, half8 *dst_group, int piece0, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NFW = (NF + WPB - 1) / WPB;
  const char *src = reinterpret_cast<const char *>(src_group) + (int64_t)piece0 * 1024 + lane * 16;
  char *dst = reinterpret_cast<char *>(dst_group) + piece0 * 1024;
#pragma unroll
  for (int i0 = 0; i0 < NFW; i0 += 4) {
    const char *s4 = src + i0 * 1024;
    auto *d4 = (__attribute__((address_space(3))) void *)(dst + i0 * 1024);
    __builtin_amdgcn_global_load_lds(s4, d4, 16, 0, 0);
    if (i0 + 1 < NFW) __builtin_amdgcn_global_load_lds(s4, d4, 16, 1024, 0);
    if (i0 + 2 < NFW) __builtin_amdgcn_global_load_lds(s4, d4, 16, 2048, 0);
    if (i0 + 3 < NFW) __builtin_amdgcn_global_load_lds(s4, d4, 16, 3072, 0);
  }
#endif
}

struct ChrTab {
  int n_chr;
  int64_t cum[32];
};

struct wcx_ctx;
int wcx_refine_launch(wcx_ctx *ctx, const double *Xr, int S, int Sp, const ChrTab &tab,
                      int64_t row_begin, int64_t n_rows, const unsigned char *searched,
                      const uint2 *sl, const int *cnt_out, const unsigned int *flags,
                      const int *perm, int k, int32_t *d_out_idx, double *d_out_dist,
                      ScreenGlobals *glob, int sl_stride = CAP);
// most shortlist entries per row the refine accepts (above: the row is flagged for the exact kernel)
constexpr int REFINE_MAX = 2048;
