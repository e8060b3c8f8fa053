// Threshold estimates for the symmetric sweep from COUNTS over the low-norm rows ("hubs").
//
// The nearest neighbours of a bin are overwhelmingly low-noise bins: at 15 kb x 500 samples 93 % of all
// (target, neighbour) pairs have their neighbour among the 1/32 of the rows with the smallest centred
// norm.  The sweep order therefore starts with a HUB region H (rows below a norm quantile, cells padded
// to tiles like the rest), and this kernel meets every target row with H only -- 1/32 of the work of a
// one-directional sweep -- to find, per row, a threshold D with AT LEAST `need` hub candidates below it:
//   phase 1  the first n1 hub tiles (visited in a scrambled order: a fair sample of H): mean and
//            standard deviation of the row's screen distances to them;
//   trials   T thresholds at the normal quantiles of the ranks need x {0.85 ... 4} among the hub
//            candidates still to come;
//   phase 2  the remaining hub tiles: per trial, how many screen distances lie below it (two vector
//            instructions per output and trial, no list, no store);
//   result   the tightest trial with >= need candidates below it.
// The count is a fact, not a probability: `need` pairs DO lie below D (same accumulator arithmetic as the
// symmetric sweep: the sweep will admit exactly these and whatever else lies below D outside H), so the
// final cut finds its k entries + filter margin whatever the data -- what depends on the data is only how
// MANY other pairs D admits (15 kb: ~1.4 need; a sampled estimate with its Poisson allowance: ~2.6 need).
// SECOND PASS over the same hub tiles (A.sl != nullptr): with the thresholds fixed, every (row, hub
// candidate) pair below its row's threshold is appended to the row's list here -- this workgroup is the
// only writer of its 128 rows' lists (register counters, plain stores) -- and the symmetric sweep skips
// that direction for streamed hub tiles: four fifths of all hits fall on hub candidates, and inside the
// sweep they cost a whole workgroup ~4 000 cycles per tile pair while the matrix pipe idles (the four
// waves meet at a barrier every iteration); here the hits are the point of the pass.
// A row without an estimate (no trial reached `need`) is flagged; if more rows than the redo path takes
// end up flagged after the sweep (data without hubs), the sampled pre-pass + a second sweep run instead
// (device-side gate, see screen_sym_path).
#pragma once
#include "wave_sort.h"
#include "wcx_common.h"
#include "screen_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int CNT_T = 8;     // trial thresholds per row
// standard normal quantile (Acklam's rational approximation, |error| < 1.2e-9 in exact arithmetic; the
// trials only need it to a per cent)
__device__ __forceinline__ float ndtri_f(float p) {
  const float a[6] = {-3.969683028665376e+01f, 2.209460984245205e+02f, -2.759285104469687e+02f,
                      1.383577518672690e+02f, -3.066479806614716e+01f, 2.506628277459239e+00f};
  const float b[5] = {-5.447609879822406e+01f, 1.615858368580409e+02f, -1.556989798598866e+02f,
                      6.680131188771972e+01f, -1.328068155288572e+01f};
  const float c[6] = {-7.784894002430293e-03f, -3.223964580411365e-01f, -2.400758277161838e+00f,
                      -2.549732539343734e+00f, 4.374664141464968e+00f, 2.938163982698783e+00f};
  const float d[4] = {7.784695709041462e-03f, 3.224671290700398e-01f, 2.445134137142996e+00f,
                      3.754408661907416e+00f};
  const float pl = 0.02425f;
  if (p < pl) {
    const float q = sqrtf(-2.f * logf(p));
    return (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) /
           ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1.f);
  }
  if (p > 1.f - pl) {
    const float q = sqrtf(-2.f * logf(1.f - p));
    return -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) /
           ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1.f);
  }
  const float q = p - 0.5f, r = q * q;
  return (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q /
         (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1.f);
}

constexpr float CNT_VALID = -2.1e9f;   // accumulators below this belong to padding / non-finite rows (theirs: <= -4.29e9;
                                       // a usable threshold: D < SYM_DMAX = 4e9, i.e. acc > -2e9)

// One workgroup per target quad (4 tiles of the sweep order, one per wave); NK, CTG, RING as in
// k_screen_sym (same fragments, same LDS-DMA ring, same accumulator: acc = -d~/2).
// APPEND = false: the counting pass (thresholds); APPEND = true: the second pass (the hub hits -> lists,
// thresholds read back from tinfo) -- two kernels, so that neither carries the other's registers.
template <int NK, int CTG, int LBW, int RING, bool APPEND>
__global__ __launch_bounds__(256, LBW) void k_screen_count(const CountArgs A) {
  constexpr int WPB = 4;
  constexpr int TILE_H8 = CTG * NK * 64;
  constexpr int NPIECE = CTG * NK;
  constexpr int NPW = (NPIECE + WPB - 1) / WPB;
  static_assert(RING >= 2 && (RING - 2) * NPW <= 63, "vmcnt range");
  extern __shared__ __align__(16) unsigned char smem[];
  half8 *sbuf = reinterpret_cast<half8 *>(smem);
  int *glist = reinterpret_cast<int *>(smem + RING * TILE_H8 * 16);
  __shared__ int s_nlist;
  if (A.gate && *A.gate) return;                         // (the other estimator's turn)
  const int n_tiles = (int)A.glob->n_tiles;
  const int quad = blockIdx.x, t0 = quad * 4;
  if (t0 >= n_tiles) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, hf = lane >> 5;
  const int t = t0 + wave;
  int ngr = ((int)A.glob->n_hub_tiles + CTG - 1) / CTG;  // hub groups
  if (ngr > A.glist_cap - 64) ngr = A.glist_cap - 64;    // (a degenerate norm distribution: cut the region)
  const int n_hub = (int)A.glob->n_hub_tiles < ngr * CTG ? (int)A.glob->n_hub_tiles : ngr * CTG;
  // (the sweep skips what the second pass covers: it must see the region as it is used here)
  if (blockIdx.x == 0 && tid == 0 && n_hub < (int)A.glob->n_hub_tiles) A.glob->n_hub_tiles = (unsigned int)n_hub;
  // visit list: the hub groups in a scrambled order (multiplicative step coprime to their number), those
  // that hold a pair for any wave of the quad; entry = group | chromosome of its tiles (5 bits each)
  if (wave == 0) {
    int tq[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) tq[w] = t0 + w < n_tiles ? (int)A.tchr[t0 + w] : 255;
    int step = (int)(0.6180339887 * ngr) | 1;
    auto gcd = [](int a, int b) { while (b) { const int r = a % b; a = b; b = r; } return a; };
    while (ngr > 1 && gcd(step, ngr) != 1) step += 2;
    if (ngr <= 1) step = 1;
    int count = 0;
    for (int i0 = 0; i0 < ngr; i0 += 64) {
      const int i = i0 + lane;
      const bool in = i < ngr;
      // (the second pass walks the tiles in their own order: the lists then name their entries in sweep
      //  order for every row alike -- what keeps the refine's gathers of neighbouring rows in step)
      const int g = in ? (APPEND ? i : (int)(((long long)i * step + quad) % ngr)) : 0;
      bool keep = false;
      int ent = g;
      if (in) {
#pragma unroll
        for (int s = 0; s < CTG; ++s) {
          const int c = g * CTG + s;
          const int cc = c < n_hub ? (int)A.tchr[c] : 255;
          ent |= (cc & 31) << (20 + 5 * s);
#pragma unroll
          for (int w = 0; w < 4; ++w) keep = keep || (cc != 255 && tq[w] != 255 && cc != tq[w]);
        }
      }
      const unsigned long long bal = __ballot(keep);
      if (keep) glist[count + __popcll(bal & ((1ull << lane) - 1ull))] = ent;
      count += __popcll(bal);
    }
    if (lane == 0) s_nlist = count;
  }
  half8 th[NK];
  const bool tvalid = t < n_tiles;
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) th[ks] = A.F[((int64_t)(tvalid ? t : 0) * NK + ks) * 64 + lane];
  if (hf) {
    const half8 x = th[NK - 1];
    half8 y = x;
    y[4] = x[6]; y[5] = x[7]; y[6] = x[4]; y[7] = x[5];
    th[NK - 1] = y;
  }
  const int rowj = tvalid ? A.perm[(int64_t)t * 32 + l32] : -1;
  const int mychr = tvalid ? ((int)A.tchr[t] & 31) : 31;
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
  // tiles this wave will meet (its own chromosome's are skipped)
  int n_act_all = 0;
  for (int q = 0; q < n_my; ++q) {
    const int cur = glist[q];
#pragma unroll
    for (int s = 0; s < CTG; ++s) {
      const int cc = (cur >> (20 + 5 * s)) & 31;
      n_act_all += ((cur & 0xfffff) * CTG + s < n_hub && cc != mychr && cc != 31) ? 1 : 0;
    }
  }
  n_act_all = __builtin_amdgcn_readfirstlane(n_act_all);
  double s1 = 0.0, s2 = 0.0;
  int nv = 0, seen = 0;
  bool counting = false;
  float thr[CNT_T];
  int cj[CNT_T];
#pragma unroll
  for (int j = 0; j < CNT_T; ++j) { thr[j] = HUGE_VALF; cj[j] = 0; }
  const int n1 = A.n1 < n_act_all / 4 ? A.n1 : n_act_all / 4;     // (tiny hub regions: a quarter of them)
  float theta = HUGE_VALF;
  int chosen = -1, cntr = 0, n_app = 0;
  uint2 *mine = APPEND ? A.sl + (int64_t)(rowj >= 0 ? rowj : 0) * A.cap2 : nullptr;
  constexpr int pass = APPEND ? 1 : 0;
  if (APPEND) theta = (tvalid && rowj >= 0) ? __uint_as_float(A.tinfo[(int64_t)t * 64 + l32]) : HUGE_VALF;
  {
#pragma unroll
  for (int q = 0; q < RING - 1; ++q)
    if (q < n_my) fetch(glist[q] & 0xfffff, q);
  for (int q = 0; q < n_my; ++q) {
    const int cur = __builtin_amdgcn_readfirstlane(glist[q]);
    const int g = cur & 0xfffff;
    const int slot = q % RING;
    const half8 *sb = sbuf + slot * TILE_H8;
    {
      const int younger = n_my - 1 - q < RING - 2 ? n_my - 1 - q : RING - 2;
      if (younger >= 2 && RING >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory");
      else if (younger == 1 && RING >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (q + RING - 1 < n_my) fetch(glist[q + RING - 1] & 0xfffff, (q + RING - 1) % RING);
    }
    bool act[CTG];
    bool any_act = false;
#pragma unroll
    for (int s = 0; s < CTG; ++s) {
      const int cc = (cur >> (20 + 5 * s)) & 31;
      act[s] = tvalid && g * CTG + s < n_hub && cc != mychr && cc != 31;
      any_act = any_act || act[s];
    }
    if (!any_act) continue;                          // wave-uniform
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
    if constexpr (APPEND) {
      // second pass: column-direction hits against the fixed threshold, straight to my row's list
#pragma unroll
      for (int s = 0; s < CTG; ++s) {
        if (!act[s]) continue;                         // wave-uniform
        float m = fmaxf(fmaxf(acc[s][0], acc[s][1]), acc[s][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) m = fmaxf(fmaxf(m, acc[s][r]), acc[s][r + 1]);
        m = fmaxf(m, acc[s][15]);
        if (!__any(m >= theta)) continue;
        unsigned int pm = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          pm = __builtin_amdgcn_alignbit(pm, ~__float_as_uint(acc[s][r] - theta), 31);
        pm &= 0xffffu;
        const unsigned int anym = wcx::wave_or_u32(pm);
        if (anym == 0) continue;
        const unsigned int pc = (unsigned int)__popc(pm);
        n_app += (int)pc;
        const auto pcs = __builtin_amdgcn_permlane32_swap(pc, pc, false, false);
        int ofs = cntr + (hf ? (int)pcs[0] : 0);
        cntr += (int)(pcs[0] + pcs[1]);
        const unsigned int cposb = (unsigned int)((g * CTG + s) * 32 + 4 * hf);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (anym & (0x8000u >> r)) {
            asm volatile("" ::: "memory");               // (keeps the two tests separate)
            if (pm & (0x8000u >> r)) {
              if (ofs < A.cap2)
                mine[ofs] = make_uint2(__float_as_uint(-2.f * acc[s][r]),
                                       cposb + (unsigned int)(8 * (r >> 2) + (r & 3)));
              ++ofs;
            }
          }
        }
      }
      continue;
    } else {
#pragma unroll
    for (int s = 0; s < CTG; ++s) {
      if (!act[s]) continue;                           // wave-uniform
      if (!counting) {
        // phase 1: moments of this row's accumulators (= -d~/2) over a fair sample of the hubs
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
          // the target's two lanes hold half of its outputs each
          const double t1 = s1 + __shfl_xor(s1, 32, 64), t2 = s2 + __shfl_xor(s2, 32, 64);
          const int tn = nv + __shfl_xor(nv, 32, 64);
          const double mu = tn > 0 ? t1 / tn : 0.0;
          double var = tn > 1 ? t2 / tn - mu * mu : 0.0;
          var = var > 0.0 ? var : 0.0;
          const float sd = (float)sqrt(var), muf = (float)mu;
          // hub candidates still to come for this row (padding rows included: a slight overestimate)
          const float n2 = 32.f * (float)(n_act_all - seen);
          const float mult[CNT_T] = {0.85f, 1.0f, 1.15f, 1.35f, 1.6f, 2.0f, 2.7f, 4.0f};
#pragma unroll
          for (int j = 0; j < CNT_T; ++j) {
            const float qf = mult[j] * (float)A.need / (n2 > 1.f ? n2 : 1.f);    // upper-tail fraction
            float th_j = CNT_VALID;                                                // everything real
            if (qf < 0.97f && tn > 8) th_j = muf - ndtri_f(qf) * sd;               // large acc = small d~
            thr[j] = th_j > CNT_VALID ? th_j : CNT_VALID;
          }
          counting = true;
        }
        continue;
      }
      // phase 2: per trial, the outputs at or above it (d~ at or below the trial distance)
#pragma unroll
      for (int j = 0; j < CNT_T; ++j) {
        int c = cj[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) c += (acc[s][r] >= thr[j]) ? 1 : 0;
        cj[j] = c;
      }
    }
    }
  }
  if constexpr (!APPEND) {
  // tightest trial with `need` hub candidates below it
#pragma unroll
  for (int j = CNT_T - 1; j >= 0; --j) {
    const int tot = cj[j] + __shfl_xor(cj[j], 32, 64);
    if (counting && tot >= A.need) { theta = thr[j]; chosen = j; }
  }
  if (!tvalid || rowj < 0 || chosen < 0 || !(theta > CNT_VALID)) theta = HUGE_VALF;
  }
  }
  (void)pass;
  if constexpr (APPEND) {
    if (tvalid && hf == 0 && rowj >= 0) {
      A.cnt[rowj] = cntr;                            // entries this pass appended
      if (cntr > A.cap2) A.flags[rowj] = 1u;
    }
    if (A.stats) {
      const int tot_a = wcx::wave_sum_i(n_app);
      if (lane == 0) atomicAdd(&A.stats[4], (unsigned long long)tot_a);
    }
    return;
  }
  if (tvalid && hf == 0) {
    const int64_t p = (int64_t)t * 32 + l32;
    if (rowj >= 0) {
      if (!(theta < HUGE_VALF)) A.flags[rowj] = 1u;
      A.Dest[rowj] = -2.f * theta;                 // threshold in screen-distance space (-inf if none)
      A.cnt[rowj] = 0;
    }
    A.tinfo[(p >> 5) * 64 + l32] = __float_as_uint(theta);
    A.tinfo[(p >> 5) * 64 + 32 + l32] = (unsigned int)rowj;
  }
  if (tvalid) {
    float m = (hf == 0) ? theta : HUGE_VALF;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {
      const float o = __shfl_xor(m, off, 64);
      m = o < m ? o : m;
    }
    if (lane == 0) A.tmin[t] = m;
    if (A.stats) {                                     // (diagnostics: mean trial chosen, rows without one)
      const bool is_row = hf == 0 && rowj >= 0;
      const int cs = wcx::wave_sum_i(is_row && chosen >= 0 ? chosen : 0), cf = wcx::wave_sum_i(is_row && chosen < 0 ? 1 : 0);
      if (lane == 0) { atomicAdd(&A.stats[16], (unsigned long long)cs); atomicAdd(&A.stats[17], (unsigned long long)cf); }
    }
  }
}

template <int NK, int CTG, int LBW, int RING>
int count_launch_t(const CountArgs &a, unsigned grid, size_t lds, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_screen_count<NK, CTG, LBW, RING, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_screen_count<NK, CTG, LBW, RING, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  if (a.append_pass) k_screen_count<NK, CTG, LBW, RING, true><<<grid, 256, lds, st>>>(a);
  else k_screen_count<NK, CTG, LBW, RING, false><<<grid, 256, lds, st>>>(a);
  return (int)hipGetLastError();
}
#define WCX_COUNT_TRY(N, C, L, R) \
  if (nk == N && ctg == C && lb == L && ring == R) return count_launch_t<N, C, L, R>(a, grid, lds, st);

}  // namespace
