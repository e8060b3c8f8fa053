// The MFMA screen kernel of the reference-bin search and its device helpers (see
// newref_topk_screen.hip for the pipeline and the error budget).  Included by the orchestration
// unit and by the instantiation units newref_screen_k*.hip (the configurations are spread over
// several translation units so that they compile in parallel).
#pragma once
#include "wave_sort.h"
#include "wcx_common.h"
#include "screen_common.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ unsigned int f32_key(float t) {
  const unsigned int u = __float_as_uint(t);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned int kx) {
  const unsigned int u = (kx & 0x80000000u) ? (kx & 0x7fffffffu) : ~kx;
  return __uint_as_float(u);
}
__device__ __forceinline__ float up(float v) {  // a float strictly above v (v >= 0, finite)
  return v * 1.0000005f + 1e-37f;
}

// Two fp16 values h1 + h2 bracketing x from below (up = false) or above (up = true), with
// sum = h1 + h2 EXACT in fp32 and no fp16 subnormals (quantum >= 2^-14; the matrix pipe may
// flush them).  |x| <= 65000.  Error |x - sum| < max(2^-21 |x|, 2^-14).
__device__ __forceinline__ void split16(float x, bool up, _Float16 &h1, _Float16 &h2, float &sum) {
  h1 = (_Float16)x;                       // round to nearest
  const float f1 = (float)h1;
  const float r = x - f1;                 // exact (Sterbenz)
  int eb = (int)((__float_as_uint(f1) >> 23) & 0xffu) - 21;
  if (eb < 113) eb = 113;
  const float q = __uint_as_float((unsigned int)eb << 23);
  const float m = up ? ceilf(r / q) : floorf(r / q);   // |m| <= 1024
  const float f2 = m * q;
  h2 = (_Float16)f2;                      // exact
  sum = f1 + f2;                          // exact: a multiple of q below 2^(e+1)
}

constexpr float AUG = 32768.f;            // the constant factor of the augmented products
constexpr float GMAX = 4.0e9f;            // thresholds at or above this count as "none yet"
constexpr float SLOW_OFF = 3.75e9f;       // slow-path pass offset: real rows pass, padding fails


// ------------------------------------------------------------------------------------------
// Error budget of one target row (see the file header): na = |a~|^2 as encoded, E, Q.
// |computed t - exact hi-plane t| <= Q: fp32 accumulation of the 16 NK products (data columns
// + the nb'/2 and G'/2 columns: sum |x y| <= N_a N_max + (N_a + N_max)^2), the rounding of
// t = G' - 2 acc, and nb - nb' < 2^-21 nb + 4.  gamma = (16 NK + 12) 2^-23 (the products, the
// augmented columns and the adds of the partial accumulator chains; any summation order obeys it).
__device__ __forceinline__ void row_budget(const RowInfo &ti, float e_max, float N_max, float gamma,
                                           float &na, float &E, float &Q) {
  na = ti.nb;
  E = up(ti.e + e_max);
  const float nsum = ti.N + N_max;
  Q = up(2.f * gamma * ti.N * N_max + 4.8e-7f * (ti.N * ti.N + 2.f * N_max * N_max) +
         2.2f * gamma * nsum * nsum + 4.f);
}
constexpr float G_INIT = 3.0e38f;   // "no threshold yet" (finite on purpose)
constexpr int CNT_MASK = 0x3fffffff;   // shortlist count; bit 30 of the stored word = "estimate"

// All CAP slots of a shortlist exist in memory: load unconditionally (16 independent loads in
// flight; a load under `if (e < n)` made the compiler wait for each one in turn -- 16 serial
// round trips to HBM, ~40k cycles per cut) and mask afterwards.
template <int NSL>
__device__ __forceinline__ void load_shortlist(const uint2 *__restrict__ sl_row, uint2 (&raw)[NSL]) {
  const int lane = wcx::lane_id();
#pragma unroll
  for (int q = 0; q < NSL; ++q) raw[q] = sl_row[q * 64 + lane];
}

// Wave-level cut of one target's shortlist (entries = float bits of t, sweep position).
//
// State of a target: its threshold G (everything with t <= G has been kept so far) and whether G
// is RIGOROUS (>= the filter bound F of the true k-th neighbour, see the file header) or an
// ESTIMATE taken from a sample of the candidates (est = 1: overwhelmingly likely large enough,
// proven or refuted at the end).
//   mode 0  in-sweep cut    kk = k:  the kk-th smallest t of the list bounds the true k-th (any k
//           actual candidates do), so Gn = F(t_kk) is rigorous -- usable iff everything with
//           t <= Gn is in the list, i.e. Gn <= G.  Resolution 2^-11 (the fp16 screen's own).
//   mode 1  end of the sampled pre-pass, kk = r << k: Gn becomes the estimate (est = 1).
//   mode 2  final cut, exact k-th key.
// fail_if_est: the last cut of a row -- an estimate that is still unproven hands the row to the
// exact kernel.  Returns the new G; n_out / est_out.
template <int NSL>
__device__ __forceinline__ float compact_loaded(const uint2 (&raw)[NSL], int n,
                                                uint2 *__restrict__ sl_row, int kk, float na, float E,
                                                float Q, float G_old, int est_old, int mode,
                                                bool fail_if_est, unsigned int *overflow_flag,
                                                int &n_out, int &est_out, bool raw_est = false) {
  const int lane = wcx::lane_id();
  unsigned int key[NSL], idx[NSL];
#pragma unroll
  for (int q = 0; q < NSL; ++q) {
    const bool in = q * 64 + lane < n;
    key[q] = in ? f32_key(__uint_as_float(raw[q].x)) : 0xffffffffu;
    idx[q] = in ? raw[q].y : 0u;
  }
  float G = G_old;
  int est = est_old;
  const bool exact = mode == 2;
  if (n >= kk) {
    // kk-th smallest key by bitwise bisection (ballot counts): largest v with #(key < v) < kk.
    // The keys share their leading bits (sign, exponent, ...): start below the common prefix.
    const unsigned int kref = (unsigned int)__builtin_amdgcn_readfirstlane((int)key[0]);
    unsigned int x = 0;
#pragma unroll
    for (int q = 0; q < NSL; ++q) x |= (q * 64 + lane < n) ? (key[q] ^ kref) : 0u;
    x = wcx::wave_or_u32(x);
    const int hb = 31 - __builtin_clz(x | 1u);              // highest differing bit (0 if none)
    unsigned int prefix = kref & ~((2u << hb) - 1u);
    const int low = exact ? 0 : 12;
    for (int bit = hb; bit >= low; --bit) {
      const unsigned int trial = prefix | (1u << bit);
      int c = 0;
#pragma unroll
      for (int q = 0; q < NSL; ++q) c += __popcll(__ballot(key[q] < trial));
      if (c < kk) prefix = trial;
    }
    if (!exact && hb >= 12) prefix |= 0xfffu;
    if (!exact && hb < 12) prefix |= (2u << hb) - 1u;       // all keys within the low bits: upper end
    const float tk = key_f32(prefix);
    // T-space -> distance space -> filter bound F -> back to t-space, rounded outwards
    float dk = tk + na;
    dk = dk > 0.f ? dk : 0.f;
    const float rt = sqrtf(up(dk + Q)) * 1.0000005f + 2.f * E;
    const float Fb = up(up(rt * rt) + Q);
    float Gn = (Fb - na) + 4e-7f * (Fb + na);
    // An ESTIMATE needs no filter margin: nothing is proven with it -- the final cut proves (k entries
    // whose own filter bound lies below the estimate) or flags the row.  Taking the r-th sample value as
    // it is admits a third fewer pairs than its filter bound (the distances of a row's nearest few
    // hundred candidates differ by a few per cent only: a 2 % wider threshold admits 1.5 x as many);
    // the sample rank r allows for the margin instead (sample_rank: the estimate must cover the rank of
    // the k-th neighbour's FILTER BOUND, ~1.14 k, not k).
    if (mode == 1 && raw_est) Gn = tk + 4e-7f * (fabsf(tk) + na);
    if (tk < HUGE_VALF) {        // (NaN / inf bound: keep the old threshold)
      if (mode == 1) { if (Gn < G_old) { G = Gn; est = 1; } }
      else if (Gn <= G_old) { G = Gn; est = 0; }
    }
  }
  // keep entries with t <= G
  const unsigned int gkey = f32_key(G);
  int base = 0;
#pragma unroll
  for (int q = 0; q < NSL; ++q) {
    const bool keep = (q * 64 + lane < n) && (key[q] <= gkey);
    const unsigned long long m = __ballot(keep);
    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
    if (keep) sl_row[pos] = make_uint2(__float_as_uint(key_f32(key[q])), idx[q]);
    base += __popcll(m);
  }
  if (base > LIM || (fail_if_est && est)) {   // no room / unproven estimate: exact kernel
    if (lane == 0) *overflow_flag = 1u;
    n_out = 0;
    est_out = 0;
    return -HUGE_VALF;
  }
  n_out = base;
  est_out = est;
  return G;
}

// Threshold -> the two fp16 values of the target's augmented columns and the value G' they encode.
__device__ __forceinline__ void encode_threshold(float G, _Float16 &w1, _Float16 &w2, float &Gp) {
  if (G < -GMAX) {            // nothing may pass (unused target lane, overflowed row)
    w1 = (_Float16)-65504.f; w2 = (_Float16)-65504.f; Gp = -131008.f * 65536.f;
  } else if (G < GMAX) {
    float s;
    split16(G * (1.f / 65536.f), true, w1, w2, s);
    Gp = s * 65536.f;
  } else {                    // no threshold yet: columns off, the slow path passes every real row
    w1 = (_Float16)0; w2 = (_Float16)0; Gp = 0.f;
  }
}

// Cuts of the targets flagged in `need` (bits = targets of one 32-target tile of this wave,
// shortlists at w_srow + bit); in a burst the next target's shortlist is loaded while the current
// one is selected and written back.  tpos / G / cntr / est are the tile's per-lane state (lanes l
// and l + 32 hold target l), th_last the last k-step of its B operand (augmented columns).
template <int NK, bool LOOKAHEAD, int NSL>
__device__ __forceinline__ void cut_targets(const ScreenArgs &A, unsigned int need, int mode,
                                            int64_t w_srow, float e_max, float N_max, float gamma,
                                            int tpos, float &G, float &Gp, int &cntr, int &est,
                                            half8 &th_last, int &n_compact) {
  const int lane = wcx::lane_id();
  const int l32 = lane & 31, hf = lane >> 5;
  const int kk = mode == 2 ? A.k : A.cut_k;
  const bool fail_if_est = (mode == 2) && A.n_seg == 1;
  // this wave's appends must be visible before they are re-read
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  int c = __ffs((int)need) - 1;
  need &= need - 1;
  uint2 raw[NSL];
  load_shortlist(A.sl + (w_srow + c) * (int64_t)CAP, raw);
  for (;;) {
    const int cn = need ? __ffs((int)need) - 1 : -1;
    need &= need - 1;
    uint2 rawn[LOOKAHEAD ? NSL : 1];
    if constexpr (LOOKAHEAD) {
      if (cn >= 0) load_shortlist(A.sl + (w_srow + cn) * (int64_t)CAP, rawn);
    }
    const RowInfo ti = A.info[__builtin_amdgcn_readlane(tpos, c)];
    float na_c, E_c, Q_c;
    row_budget(ti, e_max, N_max, gamma, na_c, E_c, Q_c);
    const float G_c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(G), c));
    const int n_c = __builtin_amdgcn_readlane(cntr, c);
    const int e_c = __builtin_amdgcn_readlane(est, c);
    int n_new, e_new;
    const float Gn = compact_loaded(raw, n_c, A.sl + (w_srow + c) * (int64_t)CAP, kk, na_c, E_c,
                                    Q_c, G_c, e_c, mode, fail_if_est, &A.flags[w_srow + c], n_new,
                                    e_new, A.raw_est != 0);
    if (l32 == c) { G = Gn; cntr = n_new; est = e_new; }
    ++n_compact;
    if (cn < 0) break;
    c = cn;
    if constexpr (LOOKAHEAD) {
#pragma unroll
      for (int q = 0; q < NSL; ++q) raw[q] = rawn[q];
    } else {                       // no registers to spare for the look-ahead
      load_shortlist(A.sl + (w_srow + c) * (int64_t)CAP, raw);
    }
  }
  _Float16 w1, w2;
  encode_threshold(G, w1, w2, Gp);
  if (hf) { th_last[6] = w1; th_last[7] = w2; }
  // nothing loaded on this (rare) path may look pending to the compiler at the top of the main
  // loop: it would park an s_waitcnt vmcnt(0) inside the MFMA block, where it also drains the
  // group prefetch -- a full L2 round trip per iteration
  __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0)
}


// NK = k-steps of 16 (K = 16 NK >= S + 4), CTG = candidate sub-tiles of 32 rows per iteration,
// TT = target tiles of 32 rows per wave (B operands resident in registers), WPB = waves per
// workgroup: a workgroup owns 32 TT WPB target rows of one chromosome.
// PROF = per-phase s_memtime accounting into stats[8..13] (diagnostics, debug flag 4).
template <int NK, int CTG, int TT, int WPB, int LBW, int RING, bool PROF = false>
__global__ __launch_bounds__(64 * WPB, LBW) void k_screen(
    const ScreenArgs A) {
  // The candidate sweep is cut into chunks of groups, one launch per chunk: every workgroup of
  // a launch streams the SAME few MB of candidate fragments, which therefore come out of the
  // XCD L2s instead of HBM/MALL.  Per-target state (threshold G, shortlist count, estimate bit)
  // lives in g_state/cnt between launches; the shortlists are in HBM anyway.
  constexpr int NTH = 64 * WPB;
  constexpr int GR = CTG * 32;                      // candidate rows per iteration
  constexpr int TILE_H8 = CTG * NK * 64;            // half8 elements per staged candidate group
  constexpr int NPT = (TILE_H8 + NTH - 1) / NTH;    // 16-byte pieces per thread
  constexpr int SH = CTG >= 2 ? 2 : 1;              // sub-tiles per epilogue half
  constexpr int NH = CTG / SH;                      // epilogue halves per iteration
  constexpr int NOUT = SH * 16;                     // screen outputs per lane, tile and half
  constexpr int WT = TT * 32;                       // target rows per wave
  constexpr bool LOOKAHEAD = (NK <= 8 && TT == 1 && LBW <= 3);  // registers to spare for the next shortlist
  // Staging of the candidate groups.  RING == 0: through registers (global_load -> ds_write) into a
  // double buffer.  RING >= 2: by LDS-DMA (global_load_lds: one wave-wide instruction lands 1 KiB =
  // one (sub-tile, k-step) fragment; the fragment order of F is the LDS order, so the copy is
  // linear) into a ring of RING slots, RING - 1 groups ahead of the matrix pipe: no staging
  // registers, no ds_write pass, and the L2 latency of a group is covered by RING - 2 whole
  // iterations.  Every wave issues the same number of pieces (NPW; a wave short of pieces repeats
  // one -- same bytes, same place) so that the counted s_waitcnt below is a literal.
  constexpr bool DMA = RING >= 2;
  constexpr int NSLOT = DMA ? RING : 2;
  constexpr int NPW = (CTG * NK + WPB - 1) / WPB;   // 1 KiB pieces per wave and group (DMA)
  static_assert(!DMA || (RING - 2) * NPW <= 63, "vmcnt range");
  extern __shared__ __align__(16) unsigned char smem[];
  half8 *sbuf = reinterpret_cast<half8 *>(smem);                       // [NSLOT][TILE_H8]
  int *glist = reinterpret_cast<int *>(smem + NSLOT * TILE_H8 * 16);   // [groups of the chunk]
  __shared__ int s_nlist;
  if (A.gate && !*A.gate) return;                   // (a second-attempt launch nobody asked for)

  // Candidate segments: with few target blocks (a row shard of a multi-GPU build) every block is
  // issued n_seg times; copy `seg` sweeps every n_seg-th group of the visit list into its own
  // shortlists / thresholds (arrays offset by seg * n_rows_all); k_merge_segments joins them.
  const int seg = (int)blockIdx.x / A.n_blocks;
  const ScreenBlock blk = A.blocks[(int)blockIdx.x - seg * A.n_blocks];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, hf = lane >> 5;
  const int64_t soff = (int64_t)seg * A.n_rows_all;
  const int64_t wg_srow = blk.row0 - A.row_begin + soff;
  uint2 *wg_sl = A.sl + wg_srow * (int64_t)CAP;    // this workgroup's shortlists (uniform base)

  // Visit list of this launch, built once per workgroup in LDS: groups holding only
  // own-chromosome rows are skipped (gmask = chromosomes present per 64 rows); bit 31 marks groups
  // that also contain own-chromosome rows.
  const unsigned int blkbit = 1u << blk.chr;
  if (wave == 0) {
    int count = 0;
    for (int i0 = 0; i0 < A.g_count; i0 += 64) {
      const int i = i0 + lane;
      const bool in = i < A.g_count;
      const int64_t g = A.g_start + i;
      unsigned int m = blkbit;
      if (in) m = A.gmask[(g * GR) >> 6];
      const bool keep = in && m != blkbit;
      const unsigned long long bal = __ballot(keep);
      if (keep) glist[count + __popcll(bal & ((1ull << lane) - 1ull))] =
          (int)g | ((m & blkbit) ? (int)0x80000000 : 0);
      count += __popcll(bal);
    }
    if (lane == 0) s_nlist = count;
  }

  // target operands (B operands of the MFMA) stay in registers for the whole sweep
  int tl[TT], tpos[TT], cntr[TT], est[TT];
  bool tvalid[TT];
  float G[TT], Gp[TT];
  half8 th[TT][NK];
  const float e_max = __uint_as_float(A.glob->e_max), N_max = __uint_as_float(A.glob->N_max);
  const float gamma = (float)(16 * NK + 12) * 1.1920929e-7f;   // products (+ partial-sum adds)
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    tl[tt] = wave * WT + tt * 32 + l32;             // local target of this lane
    tvalid[tt] = tl[tt] < blk.nrows;
    const int64_t trow = blk.row0 + (tvalid[tt] ? tl[tt] : 0);
    tpos[tt] = A.rowpos[trow];                      // sweep position of the target row
    const int64_t ttile = tpos[tt] >> 5;
    const int trl = tpos[tt] & 31;
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) th[tt][ks] = A.Ft[(ttile * NK + ks) * 64 + trl + 32 * hf];
    const int c0 = (A.first || !tvalid[tt]) ? 0 : A.cnt[wg_srow + tl[tt]];
    cntr[tt] = c0 & CNT_MASK;     // identical in the target's two lanes
    est[tt] = (c0 >> 30) & 1;
    G[tt] = tvalid[tt] ? (A.first ? G_INIT : A.g_state[wg_srow + tl[tt]]) : -HUGE_VALF;
    _Float16 w1, w2;
    encode_threshold(G[tt], w1, w2, Gp[tt]);
    if (hf) { th[tt][NK - 1][4] = (_Float16)AUG; th[tt][NK - 1][5] = (_Float16)AUG;
              th[tt][NK - 1][6] = w1; th[tt][NK - 1][7] = w2; }
  }
  constexpr int SMALL_NSL = 5;
  const bool small_cut = A.trig + 64 <= SMALL_NSL * 64;     // wave-uniform (a kernel argument)
  int n_compact = 0, n_app = 0;
  unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, tp = 0;
  auto stamp = [&](int ph) {
    if (PROF) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      pt[ph] += now - tp;
      tp = now;
    }
  };

  half8 pre[DMA ? 1 : NPT];
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int piece0 = stage_piece0<CTG * NK, WPB>(wave_u);
  auto fetch = [&](int gix, int slot) {
    const half8 *src = A.F + (int64_t)gix * TILE_H8;
    if constexpr (DMA) {
      stage_group<CTG * NK, WPB>(src, sbuf + slot * TILE_H8, piece0, lane);
    } else {
#pragma unroll
      for (int p = 0; p < NPT; ++p)
        if ((p + 1) * NTH <= TILE_H8 || p * NTH + tid < TILE_H8) pre[p] = src[p * NTH + tid];
    }
  };
  auto park = [&](int slot) {   // staged registers -> LDS (nothing to do with DMA)
    if constexpr (!DMA) {
      half8 *so = sbuf + slot * TILE_H8;
#pragma unroll
      for (int p = 0; p < NPT; ++p)
        if ((p + 1) * NTH <= TILE_H8 || p * NTH + tid < TILE_H8) so[p * NTH + tid] = pre[p];
    }
  };
  // All loads so far (target fragments, per-target state) must have landed BEFORE the loop: left to
  // itself the compiler parks their s_waitcnt vmcnt(0) at the first use inside the loop body, where
  // it also drains the group prefetch just issued -- every iteration then pays a full L2 round
  // trip (this alone held the sweep at about a third of the matrix-pipe rate).
  __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) asm volatile("" : "+v"(th[tt][ks]));   // (loads are history)
#endif
  __syncthreads();
  const int n_list = s_nlist;
  // this segment's share of the visit list: entries seg, seg + n_seg, ...
  const int n_my = n_list > seg ? (n_list - seg + A.n_seg - 1) / A.n_seg : 0;
  auto entry = [&](int q) { return glist[seg + q * A.n_seg]; };
  if constexpr (DMA) {
#pragma unroll
    for (int q = 0; q < RING - 1; ++q)
      if (q < n_my) fetch(entry(q) & 0x7fffffff, q);
  } else {
    if (n_my > 0) {
      fetch(entry(0) & 0x7fffffff, 0);
      park(0);
    }
    __syncthreads();
  }
  bool fast = false;

  for (int q = 0; q < n_my; ++q) {
    // Register staging: issue the next group's global loads, run the MFMA block on the current
    // LDS buffer, park the loaded group in the other buffer, barrier, THEN do the shortlist
    // appends (their stores share the vmcnt counter with the loads; in this order nobody waits
    // for a store until a whole MFMA block later).
    // DMA ring: wait until this group has landed (all but the younger groups' pieces), barrier
    // (everybody's pieces landed; everybody is done with the slot refilled next), issue the
    // group RING - 1 ahead, MFMA block, appends.
    const int cur = entry(q);
    const bool more = q + 1 < n_my;
    int slot = q & 1;
    if constexpr (DMA) slot = q % RING;
    half8 *sb = sbuf + slot * TILE_H8;
    const int gix = cur & 0x7fffffff;
    const bool mixed = cur < 0;                  // some own-chromosome rows in this group
    if (PROF) tp = __builtin_amdgcn_s_memtime();
    if constexpr (DMA) {
      const int younger = n_my - 1 - q < RING - 2 ? n_my - 1 - q : RING - 2;
      if (younger >= 2 && RING >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory");
      else if (younger == 1 && RING >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp(1);                                  // wait for the group
      __builtin_amdgcn_s_barrier();
      stamp(2);                                  // barrier
      if (q + RING - 1 < n_my) fetch(entry(q + RING - 1) & 0x7fffffff, (q + RING - 1) % NSLOT);
    } else {
      if (more) fetch(entry(q + 1) & 0x7fffffff, slot ^ 1);
    }

    // acc = g~ - nb'/2 + G'/2 straight out of the matrix pipe (see k_screen_prep); A fragments
    // are read lane-linearly (conflict-free ds_read_b128), a few reads ahead of their MFMAs, and
    // every fragment feeds the TT resident target tiles.
    // PC partial accumulator chains per output tile (k-step ks feeds partial ks % PC, added up after
    // the block).  Measured (scripts/ubench/mfma_lds.hip): back-to-back MFMAs on ONE accumulator
    // already run at the full issue rate on gfx950, so PC = 1; the knob stays for experiments.
    constexpr int PC = 1;
    f32x16 acc[TT][CTG];
    {
      f32x16 part[TT][CTG][PC];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int sub = 0; sub < CTG; ++sub)
#pragma unroll
          for (int p = 0; p < PC; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[tt][sub][p][r] = 0.f;
      half8 a[NK][CTG];
#pragma unroll
      for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int sub = 0; sub < CTG; ++sub) a[ks][sub] = sb[(sub * NK + ks) * 64 + lane];
#pragma unroll
      for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int sub = 0; sub < CTG; ++sub)
#pragma unroll
          for (int tt = 0; tt < TT; ++tt)
            part[tt][sub][ks % PC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                a[ks][sub], th[tt][ks], part[tt][sub][ks % PC], 0, 0, 0);
      // schedule: PRE reads up front, then one read per TT MFMAs, the last PRE*TT MFMAs back to back
      constexpr int NR = NK * CTG, PRE0 = TT > 1 ? 4 : 6, PRE = NR < PRE0 ? NR : PRE0;
      __builtin_amdgcn_sched_group_barrier(0x100, PRE, 0);
#pragma unroll
      for (int i = 0; i < NR - PRE; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, TT, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, PRE * TT, 0);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int sub = 0; sub < CTG; ++sub) {
          acc[tt][sub] = part[tt][sub][0];
#pragma unroll
          for (int p = 1; p < PC; ++p) acc[tt][sub] += part[tt][sub][p];
        }
    }
    stamp(0);                                    // loads issued + MFMA block issued
    if constexpr (!DMA) {
      if (more) park(slot ^ 1);
      stamp(1);                                  // wait for the loads + LDS writes
      __syncthreads();
      stamp(2);                                  // barrier
    }
    // C[row = candidate][col = target].  The epilogue handles the sub-tiles in halves of SH <= 2:
    // output rr = s*16 + r of a half is candidate row
    // loc(rr) = (h*SH + s)*32 + 8*(r>>2) + 4*(lane>>5) + (r&3) of this group.  Bit (31-rr) of pmask.
    if (A.dbg & 2) {   // (diagnostics: matrix pipe + staging only, no epilogue; results invalid)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int sub = 0; sub < CTG; ++sub) {
#if defined(__HIP_DEVICE_COMPILE__)
          asm volatile("" ::"v"(acc[tt][sub]));
#endif
        }
      continue;
    }
    if (!fast) {
      bool all_set = true;
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) all_set = all_set && (G[tt] < GMAX);
      fast = __all(all_set);                     // G only ever decreases
    }
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      unsigned int pmask[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        unsigned int negs[SH];
        if (fast) {
#pragma unroll
          for (int sub = 0; sub < SH; ++sub) {   // two short dependent chains per sub-tile
            unsigned int lo8 = 0, hi8 = 0;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              hi8 = __builtin_amdgcn_alignbit(hi8, __float_as_uint(acc[tt][h * SH + sub][r]), 31);
              lo8 = __builtin_amdgcn_alignbit(lo8, __float_as_uint(acc[tt][h * SH + sub][8 + r]), 31);
            }
            negs[sub] = (hi8 << 8) | (lo8 & 0xffu);
          }
        } else {
          asm volatile("; slow path" ::: "memory");
          const float off = (G[tt] < GMAX) ? 0.f : SLOW_OFF;   // no threshold yet: pass real rows
#pragma unroll
          for (int sub = 0; sub < SH; ++sub) {
            negs[sub] = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r)
              negs[sub] = __builtin_amdgcn_alignbit(
                  negs[sub], __float_as_uint(acc[tt][h * SH + sub][r] + off), 31);
          }
        }
        unsigned int neg = negs[0];
        if (SH == 2) neg = (negs[0] << 16) | (negs[SH - 1] & 0xffffu);
        pmask[tt] = (~neg) << (32 - NOUT);
      }
      if (mixed) {   // rare: mask the own-chromosome rows of a mixed group
        asm volatile("; mixed group" ::: "memory");   // keep this a branch (no if-conversion)
        const int cs32 = (int)blk.cs, ce32 = (int)blk.ce;
        unsigned int ownm = 0;
#pragma unroll 1
        for (int rr = 0; rr < NOUT; ++rr) {
          const int loc = (h * SH + (rr >> 4)) * 32 + 8 * ((rr >> 2) & 3) + 4 * hf + (rr & 3);
          const int g = A.perm[(int64_t)gix * GR + loc];
          if (g >= cs32 && g < ce32) ownm |= 0x80000000u >> rr;
        }
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) pmask[tt] &= ~ownm;
        __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): see cut_targets
      }
      unsigned int anym[TT];
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        if (A.dbg & 1) pmask[tt] = 0;
        anym[tt] = wcx::wave_or_u32(pmask[tt]);  // wave-uniform
      }
      if (h == NH - 1) stamp(3);                 // MFMA completion + sign bits + OR
      const unsigned int pbase = (unsigned int)(gix * GR + h * SH * 32 + 4 * hf);
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        if (anym[tt]) {
          // slot reservation without LDS: the target's two lanes (l, l+32) swap their pass counts
          const unsigned int pc = (unsigned int)__popc(pmask[tt]);
          const auto pcs = __builtin_amdgcn_permlane32_swap(pc, pc, false, false);   // {low, high} lane's
          unsigned int ofs = (unsigned int)(tl[tt] * CAP + cntr[tt] + (hf ? (int)pcs[0] : 0));
          cntr[tt] += (int)(pcs[0] + pcs[1]);
          n_app += (int)pc;
          // cntr <= LIM + 64 = CAP: the slots exist (counts are cut back to <= LIM below)
#pragma unroll
          for (int rr = 0; rr < NOUT; ++rr) {
            if (anym[tt] & (0x80000000u >> rr)) {            // scalar branch: skip empty outputs
              asm volatile("" ::: "memory");                 // (keeps the two tests separate)
              if (pmask[tt] & (0x80000000u >> rr)) {
                const int loc0 = (rr >> 4) * 32 + 8 * ((rr >> 2) & 3) + (rr & 3);
                const float t = fmaf(-2.f, acc[tt][h * SH + (rr >> 4)][rr & 15], Gp[tt]);
                wg_sl[ofs] = make_uint2(__float_as_uint(t), pbase + loc0);
                ++ofs;
              }
            }
          }
        }
      }
      if (h == NH - 1) stamp(4);                 // appends
      // shortlist maintenance: wave-private (this wave's targets); counts only change here
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        if (anym[tt]) {
          const unsigned int need = (unsigned int)__ballot(tvalid[tt] && cntr[tt] > A.trig);  // low half
          if (need) {
            // (the sampled pre-pass cuts at r << k: its lists never exceed trig + 64 entries -- five
            //  64-entry slices instead of sixteen to load, select and write back)
            if (small_cut)
              cut_targets<NK, LOOKAHEAD, SMALL_NSL>(A, need, A.cut_mode, wg_srow + wave * WT + tt * 32, e_max,
                                                    N_max, gamma, tpos[tt], G[tt], Gp[tt], cntr[tt], est[tt],
                                                    th[tt][NK - 1], n_compact);
            else
              cut_targets<NK, LOOKAHEAD, CAP / 64>(A, need, A.cut_mode, wg_srow + wave * WT + tt * 32, e_max,
                                                   N_max, gamma, tpos[tt], G[tt], Gp[tt], cntr[tt], est[tt],
                                                   th[tt][NK - 1], n_compact);
          }
        }
      }
    }
    stamp(5);                                    // maintenance (cuts)
  }
  if (A.end_cut) {
    // end of the sampled pre-pass: threshold estimate from the cut_k-th smallest of the sample;
    // end of the sweep: final cut of every target's shortlist (exact k-th key)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int nv = blk.nrows - (wave * WT + tt * 32);
      if (nv > 0) {
        const unsigned int all = nv >= 32 ? 0xffffffffu : ((1u << nv) - 1u);
        if (small_cut)
          cut_targets<NK, LOOKAHEAD, SMALL_NSL>(A, all, A.end_cut, wg_srow + wave * WT + tt * 32, e_max, N_max,
                                                gamma, tpos[tt], G[tt], Gp[tt], cntr[tt], est[tt],
                                                th[tt][NK - 1], n_compact);
        else
          cut_targets<NK, LOOKAHEAD, CAP / 64>(A, all, A.end_cut, wg_srow + wave * WT + tt * 32, e_max, N_max,
                                               gamma, tpos[tt], G[tt], Gp[tt], cntr[tt], est[tt],
                                               th[tt][NK - 1], n_compact);
      }
    }
  }
#pragma unroll
  for (int tt = 0; tt < TT; ++tt)
    if (tvalid[tt] && hf == 0) {
      A.g_state[wg_srow + tl[tt]] = G[tt];
      A.cnt[wg_srow + tl[tt]] = cntr[tt] | (est[tt] << 30);
    }
  if (A.stats) {
    const int tot_c = wcx::wave_sum_i(n_compact), tot_a = wcx::wave_sum_i(n_app);
    if (lane == 0) {
      atomicAdd(&A.stats[2], (unsigned long long)(tot_c / 64));
      atomicAdd(&A.stats[4], (unsigned long long)tot_a);
      if (PROF)
        for (int i = 0; i < 6; ++i) atomicAdd(&A.stats[8 + i], pt[i]);
    }
  }
}

template <int NK, int CTG, int TT, int WPB, int LBW, int RING, bool PROF>
int screen_launch_t(const ScreenArgs &a, unsigned grid, size_t lds, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute(
      reinterpret_cast<const void *>(k_screen<NK, CTG, TT, WPB, LBW, RING, PROF>),
      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  k_screen<NK, CTG, TT, WPB, LBW, RING, PROF><<<grid, 64 * WPB, lds, st>>>(a);
  return (int)hipGetLastError();
}
// (K steps, candidate sub-tiles, target tiles per wave, waves per workgroup, waves per SIMD,
//  LDS-DMA ring slots or 0, prof)
#define WCX_SCREEN_TRY(N, C, T, W, L, R, P)                                                   \
  if (c.nk == N && c.ctg == C && c.tt == T && c.wpb == W && c.lb == L && c.ring == R &&       \
      c.prof == (P ? 1 : 0))                                                                  \
    return screen_launch_t<N, C, T, W, L, R, P>(a, grid, lds, st);

}  // namespace
