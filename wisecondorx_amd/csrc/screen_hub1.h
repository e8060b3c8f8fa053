// Threshold estimates for the ONE-DIRECTIONAL sweep (screen_kernel.h) from COUNTS over the low-norm rows.
//
// Round 5 gave the symmetric sweep its thresholds from counts over the hub region (screen_count.h: a row's
// nearest neighbours are overwhelmingly the rows of smallest centred norm, so the need-th smallest screen
// distance to that region alone bounds the k-th distance overall, tightly).  This is the same estimator for
// the sweep that row shards, gonosomal passes and small-K problems take: target rows = the workgroups'
// ScreenBlocks (<= 128 rows of one chromosome, fragments in registers exactly as in k_screen), candidates =
// the HUB REGION = the head of the sweep order (the rows at or below the norm quantile glob->hub_key; the
// order is [hub rows | the rest], each best-first by (norm bucket, chromosome)), streamed through the
// LDS-DMA ring in a scrambled order:
//   phase 1  the first n1 hub tiles a wave meets: mean and standard deviation of acc = g~ - nb'/2
//            (= -t/2, t = nb' - 2 g~ the sweep's own screen value; the threshold columns are zero here)
//   trials   T thresholds at the normal quantiles of the ranks need x {..} among the hub candidates to come
//   phase 2  the other hub tiles: per trial, how many outputs lie at or above it; no list, no store
//   result   the tightest trial with >= need candidates -> g_state[row] = G = -2 thr, estimate bit set.
// The sweep then runs over ALL groups with these thresholds in its MFMA operand from the first group on
// (ScreenArgs::first = 0): no sampled phase A, ~1.4 need appends per row instead of the streaming top-k's
// k ln(B / k) or the sampled estimate's ~4 k, one cut per row (the final one, which PROVES the estimate --
// k entries whose filter bound lies below it -- or flags the row for the exact kernel, as ever).
// A row without an estimate (no trial reached need: tiny hub region, NaN row) starts the sweep without a
// threshold, i.e. as the streaming top-k it always was; loose estimates (data without hubs) overflow into
// the in-sweep cuts, which are rigorous.  Nothing here can make a result wrong, only slow.
#pragma once
#include "screen_kernel.h"
#include "screen_count.h"

#pragma clang fp contract(off)

struct Hub1Args {
  const half8 *F;              // fragments in sweep order; hub region = the first glob->n_hub_tiles tiles
  const ScreenGlobals *glob;
  const int *perm, *rowpos;
  const unsigned int *gmask;
  const ScreenBlock *blocks;
  float *g_state;              // out [seg][row - row_begin]: threshold in t-space (G_INIT = none)
  int *cnt;                    // out: estimate bit, 0 entries
  unsigned long long *stats;
  int64_t row_begin, n_rows_all;
  int n_seg;                   // copies of the per-row state to initialise (candidate segments)
  int need;                    // hub candidates wanted below the estimate
  int n1;                      // hub tiles of the moment phase
  int glist_cap;               // hub groups the visit list in LDS has room for (+ 64)
};
int wcx_hub1_launch_k1(int nk, int ctg, int lb, int ring, int trials, const Hub1Args &a, unsigned grid, size_t lds, hipStream_t st);
int wcx_hub1_launch_k2(int nk, int ctg, int lb, int ring, int trials, const Hub1Args &a, unsigned grid, size_t lds, hipStream_t st);
int wcx_hub1_launch_k3(int nk, int ctg, int lb, int ring, int trials, const Hub1Args &a, unsigned grid, size_t lds, hipStream_t st);
int wcx_hub1_launch_k4(int nk, int ctg, int lb, int ring, int trials, const Hub1Args &a, unsigned grid, size_t lds, hipStream_t st);

namespace {

template <int T> struct Hub1Mult;
template <> struct Hub1Mult<8> { static constexpr float v[8] = {0.85f, 1.0f, 1.15f, 1.35f, 1.6f, 2.0f, 2.7f, 4.0f}; };
template <> struct Hub1Mult<4> { static constexpr float v[4] = {0.9f, 1.15f, 1.6f, 3.0f}; };

template <int NK, int CTG, int LBW, int RING, int T>
__global__ __launch_bounds__(256, LBW) void k_screen_hub1(const Hub1Args A) {
  constexpr int WPB = 4;
  constexpr int GR = CTG * 32;
  constexpr int TILE_H8 = CTG * NK * 64;
  constexpr int NPIECE = CTG * NK;
  constexpr int NPW = (NPIECE + WPB - 1) / WPB;
  static_assert(RING >= 2 && (RING - 2) * NPW <= 63, "vmcnt range");
  extern __shared__ __align__(16) unsigned char smem[];
  half8 *sbuf = reinterpret_cast<half8 *>(smem);
  int *glist = reinterpret_cast<int *>(smem + RING * TILE_H8 * 16);
  __shared__ int s_nlist;
  const ScreenBlock blk = A.blocks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, hf = lane >> 5;
  const int64_t wg_srow = blk.row0 - A.row_begin;
  int ngr = (int)(((int64_t)A.glob->n_hub_tiles * 32) / GR);      // hub groups (the region ends on a 64-row border)
  if (ngr > A.glist_cap - 64) ngr = A.glist_cap - 64;               // (a degenerate norm distribution: its head)
  const unsigned int blkbit = 1u << blk.chr;
  // visit list: the hub groups in a scrambled order (multiplicative step coprime to their number, rotated
  // per workgroup: the first groups met are a fair sample of the region); groups of own-chromosome rows
  // only are skipped, bit 31 marks groups that also hold own-chromosome rows
  if (wave == 0) {
    int step = (int)(0.6180339887 * ngr) | 1;
    auto gcd = [](int a, int b) { while (b) { const int r = a % b; a = b; b = r; } return a; };
    while (ngr > 1 && gcd(step, ngr) != 1) step += 2;
    if (ngr <= 1) step = 1;
    int count = 0;
    for (int i0 = 0; i0 < ngr; i0 += 64) {
      const int i = i0 + lane;
      const bool in = i < ngr;
      const int g = in ? (int)(((long long)i * step + (long long)blockIdx.x) % ngr) : 0;
      unsigned int m = blkbit;
      if (in) m = A.gmask[((int64_t)g * GR) >> 6];
      const bool keep = in && m != blkbit && m != 0u;
      const unsigned long long bal = __ballot(keep);
      if (keep) glist[count + __popcll(bal & ((1ull << lane) - 1ull))] = g | ((m & blkbit) ? (int)0x80000000 : 0);
      count += __popcll(bal);
    }
    if (lane == 0) s_nlist = count;
  }
  // target operands exactly as in k_screen, threshold columns zero: acc = g~ - nb'/2
  const int tl = wave * 32 + l32;
  const bool tvalid = tl < blk.nrows;
  const bool wave_on = wave * 32 < blk.nrows;            // (wave-uniform)
  half8 th[NK];
  {
    const int64_t trow = blk.row0 + (tvalid ? tl : 0);
    const int tpos = A.rowpos[trow];
    const int64_t ttile = tpos >> 5;
    const int trl = tpos & 31;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) th[ks] = A.F[(ttile * NK + ks) * 64 + trl + 32 * hf];
    if (hf) { th[NK - 1][4] = (_Float16)AUG; th[NK - 1][5] = (_Float16)AUG;
              th[NK - 1][6] = (_Float16)0; th[NK - 1][7] = (_Float16)0; }
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int piece0 = stage_piece0<NPIECE, WPB>(wave_u);
  auto fetch = [&](int g, int slot) {
    stage_group<NPIECE, WPB>(A.F + (int64_t)g * TILE_H8, sbuf + slot * TILE_H8, piece0, lane);
  };
  __builtin_amdgcn_s_waitcnt(0x0F70);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) asm volatile("" : "+v"(th[ks]));
#endif
  __syncthreads();
  const int n_my = s_nlist;
  const int n_act_all = n_my * CTG;                      // tiles this workgroup's rows will meet
  double s1 = 0.0, s2 = 0.0;
  int nv = 0, seen = 0;
  bool counting = false;
  float thr[T];
  int cj[T];
#pragma unroll
  for (int j = 0; j < T; ++j) { thr[j] = HUGE_VALF; cj[j] = 0; }
  const int n1 = A.n1 < n_act_all / 4 ? A.n1 : n_act_all / 4;     // (tiny hub regions: a quarter of them)
#pragma unroll
  for (int q = 0; q < RING - 1; ++q)
    if (q < n_my) fetch(glist[q] & 0x7fffffff, q);
  for (int q = 0; q < n_my; ++q) {
    const int cur = __builtin_amdgcn_readfirstlane(glist[q]);
    const int g = cur & 0x7fffffff;
    const bool mixed = cur < 0;
    const int slot = q % RING;
    const half8 *sb = sbuf + slot * TILE_H8;
    {
      const int younger = n_my - 1 - q < RING - 2 ? n_my - 1 - q : RING - 2;
      if (younger >= 2 && RING >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory");
      else if (younger == 1 && RING >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (q + RING - 1 < n_my) fetch(glist[q + RING - 1] & 0x7fffffff, (q + RING - 1) % RING);
    }
    if (!wave_on) continue;                              // (a short last block: this wave has no rows)
    f32x16 acc[CTG];
    {
#pragma unroll
      for (int s = 0; s < CTG; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
      half8 a[NK][CTG];
#pragma unroll
      for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int s = 0; s < CTG; ++s) a[ks][s] = sb[(s * NK + ks) * 64 + lane];
#pragma unroll
      for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int s = 0; s < CTG; ++s)
          acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][s], th[ks], acc[s], 0, 0, 0);
      constexpr int NR = NK * CTG, PRE = NR < 6 ? NR : 6;
      __builtin_amdgcn_sched_group_barrier(0x100, PRE, 0);
#pragma unroll
      for (int i = 0; i < NR - PRE; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, PRE, 0);
    }
    if (mixed) {   // rare (cell borders): own-chromosome rows of the group do not count
      asm volatile("; mixed hub group" ::: "memory");
      const int cs32 = (int)blk.cs, ce32 = (int)blk.ce;
#pragma unroll
      for (int s = 0; s < CTG; ++s)
#pragma unroll 1
        for (int r = 0; r < 16; ++r) {
          const int loc = s * 32 + 8 * (r >> 2) + 4 * hf + (r & 3);
          const int row = A.perm[(int64_t)g * GR + loc];
          if (row >= cs32 && row < ce32) {
            // (a runtime index into the accumulator would send it to scratch: a select per slot instead)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) acc[s][rr] = rr == r ? -HUGE_VALF : acc[s][rr];
          }
        }
      __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): see cut_targets
    }
#pragma unroll
    for (int s = 0; s < CTG; ++s) {
      if (!counting) {
        // phase 1: moments of this row's accumulators over a fair sample of the hubs
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[s][r];
          const bool ok = v > CNT_VALID;
          const double dv = ok ? (double)v : 0.0;
          s1 += dv;
          s2 += dv * dv;
          nv += ok ? 1 : 0;
        }
        ++seen;
        if (seen >= n1) {
          const double t1 = s1 + __shfl_xor(s1, 32, 64), t2 = s2 + __shfl_xor(s2, 32, 64);
          const int tn = nv + __shfl_xor(nv, 32, 64);
          const double mu = tn > 0 ? t1 / tn : 0.0;
          double var = tn > 1 ? t2 / tn - mu * mu : 0.0;
          var = var > 0.0 ? var : 0.0;
          const float sd = (float)sqrt(var), muf = (float)mu;
          const float n2 = 32.f * (float)(n_act_all - seen);     // hub candidates still to come
#pragma unroll
          for (int j = 0; j < T; ++j) {
            const float qf = Hub1Mult<T>::v[j] * (float)A.need / (n2 > 1.f ? n2 : 1.f);   // upper-tail fraction
            float th_j = CNT_VALID;                                                     // everything real
            if (qf < 0.97f && tn > 8) th_j = muf - ndtri_f(qf) * sd;                    // large acc = small t
            thr[j] = th_j > CNT_VALID ? th_j : CNT_VALID;
          }
          counting = true;
        }
        continue;
      }
      // phase 2: per trial, the outputs at or above it
#pragma unroll
      for (int j = 0; j < T; ++j) {
        int c = cj[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) c += (acc[s][r] >= thr[j]) ? 1 : 0;
        cj[j] = c;
      }
    }
  }
  // tightest trial with `need` hub candidates at or above it
  float theta = HUGE_VALF;
  int chosen = -1;
#pragma unroll
  for (int j = T - 1; j >= 0; --j) {
    const int tot = cj[j] + __shfl_xor(cj[j], 32, 64);
    if (counting && tot >= A.need) { theta = thr[j]; chosen = j; }
  }
  float G = -2.f * theta;                                // t <= G  <=>  acc >= theta
  const bool have = tvalid && chosen >= 0 && theta > CNT_VALID && G < GMAX && G > -GMAX;
  if (tvalid && hf == 0) {
    for (int sgm = 0; sgm < A.n_seg; ++sgm) {
      const int64_t i = (int64_t)sgm * A.n_rows_all + wg_srow + tl;
      A.g_state[i] = have ? G : G_INIT;
      A.cnt[i] = have ? (1 << 30) : 0;
    }
  }
  if (A.stats && wave_on) {
    const bool is_row = hf == 0 && tvalid;
    const int cs = wcx::wave_sum_i(is_row && have ? chosen : 0), cf = wcx::wave_sum_i(is_row && !have ? 1 : 0);
    if (lane == 0) { atomicAdd(&A.stats[16], (unsigned long long)cs); atomicAdd(&A.stats[17], (unsigned long long)cf); }
  }
}

template <int NK, int CTG, int LBW, int RING, int T>
int hub1_launch_t(const Hub1Args &a, unsigned grid, size_t lds, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_screen_hub1<NK, CTG, LBW, RING, T>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  k_screen_hub1<NK, CTG, LBW, RING, T><<<grid, 256, lds, st>>>(a);
  return (int)hipGetLastError();
}
#define WCX_HUB1_TRY(N, C, L, R, TR) \
  if (nk == N && ctg == C && lb == L && ring == R && trials == TR) return hub1_launch_t<N, C, L, R, TR>(a, grid, lds, st);

}  // namespace
