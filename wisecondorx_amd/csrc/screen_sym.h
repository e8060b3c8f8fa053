// Symmetric MFMA screen of the reference-bin search (newref_tools.py:255-278): the Gram tile of a
// pair of 32-row tiles {a, b} is computed ONCE and serves both directions -- (target in b,
// candidate in a) and (target in a, candidate in b) -- which halves the matrix work of the sweep.
//
// Conditions (see screen_sym_path in newref_topk_screen.hip): every row of the matrix is a target,
// and every row already has a threshold D (screen-distance space) from the sampled pre-pass.  The
// thresholds are FIXED during this sweep, so no shortlist is ever cut here and the order of the
// candidates is free: sweep positions are sorted by (coarse norm class, chromosome) with every cell
// padded to whole tiles -- tiles are chromosome-pure (a same-chromosome tile pair is skipped
// wholesale) and low tiles hold the low-norm rows, the "hubs" most rows choose as neighbours.
//
// Work assignment: tile pair {a, b}, a < b, belongs to the wave that owns tile b (fragments in
// registers, B operand of the MFMA) and streams tile a through LDS (A operand).  The hub tiles are
// therefore always on the streamed side and their many hits arrive in the direction in which a lane
// appends to ITS OWN row's list (one atomic per lane and tile pair).
//
// The accumulator is the screen distance itself: the four augmented k-columns of the fragments
// carry both rows' quantised squared norms, acc = g~ - nb'_a/2 - nb'_b/2 = -d~/2, and a pair is
// admitted for row x iff d~ <= D_x, i.e. acc >= theta_x = -D_x/2 (exact in fp32).  Admitted pairs
// are appended to the row's list (float bits of d~, partner sweep position) through a device-scope
// atomic counter; the final cut (k_sym_final) proves the estimate or flags the row for the exact
// kernel, exactly as the one-directional sweep does.
#pragma once
#include "wave_sort.h"
#include "wcx_common.h"
#include "screen_common.h"

#pragma clang fp contract(off)

namespace {

struct SymCounters { unsigned int n_app, n_row, n_slow, n_cg, n_rg, n_ce, n_re; };

// ONE work item: target quad `quad` (4 tiles, one per wave) against the streamed tiles of chunk
// [c0, c1) that lie below its tiles -- share `split` of `n_split`.  excl: this item is the only
// writer of its rows' counters while it runs (see k_screen_sym).
// NK = k-steps of 16, CTG = streamed tiles per iteration, RING = LDS-DMA ring slots.
template <int NK, int CTG, int RING>
__device__ __forceinline__ void sym_item(const SymArgs &A, unsigned char *smem, int *s_nlist_p, int quad,
                                         int split, int n_split, int c0, int c1, int excl,
                                         SymCounters &C) {
  constexpr int WPB = 4;
  constexpr int TILE_H8 = CTG * NK * 64;             // half8 elements per streamed group
  constexpr int NPW = (CTG * NK + WPB - 1) / WPB + 1;   // DMA pieces per wave and group: fragments + one tile info
  constexpr int STG = 64;                            // staged records per wave
  static_assert(RING >= 2 && (RING - 2) * NPW <= 63, "vmcnt range");
  half8 *sbuf = reinterpret_cast<half8 *>(smem);                                   // [RING][TILE_H8]
  unsigned int *tinf = reinterpret_cast<unsigned int *>(smem + RING * TILE_H8 * 16);   // [RING][CTG][64]
  int *glist = reinterpret_cast<int *>(tinf + RING * CTG * 64);
  int &s_nlist = *s_nlist_p;

  const int n_tiles = (int)A.glob->n_tiles;
  // hub candidates' hits are already in the lists (second pass of k_screen_count): a streamed hub tile
  // is only met in the ROW direction (its rows as targets), a pair of two hub tiles not at all
  const int n_hub = A.hub_appended ? (int)A.glob->n_hub_tiles : 0;
  const int t0 = quad * 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, hf = lane >> 5;
  const int t = t0 + wave;                            // this wave's target tile
  // this work item's share of the chunk's candidate groups, below the quad's highest tile
  const int ng = (c1 - c0) / CTG;
  const int per = (ng + n_split - 1) / n_split;
  const int g_lo = c0 / CTG + split * per;
  int g_hi = g_lo + per < c1 / CTG ? g_lo + per : c1 / CTG;
  {
    const int top = t0 + 3 < n_tiles ? t0 + 3 : n_tiles;     // groups whose first tile is < top
    const int lim = (top + CTG - 1) / CTG;
    if (g_hi > lim) g_hi = lim;
  }
  if (g_lo >= g_hi) return;

  // Visit list (wave 0): a group is skipped when no wave of the quad has a pair in it -- every
  // (streamed tile c, target tile t_w) has c >= t_w or the same chromosome.  Entry = group |
  // chromosome of its tiles (5 bits each).
  if (wave == 0) {
    int tq[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) tq[w] = t0 + w < n_tiles ? (int)A.tchr[t0 + w] : 255;
    int count = 0;
    for (int i0 = g_lo; i0 < g_hi; i0 += 64) {
      const int g = i0 + lane;
      const bool in = g < g_hi;
      bool keep = false;
      int ent = g;
      if (in) {
#pragma unroll
        for (int s = 0; s < CTG; ++s) {
          const int c = g * CTG + s;
          const int cc = c < n_tiles ? (int)A.tchr[c] : 255;
          ent |= (cc & 31) << (20 + 5 * s);
#pragma unroll
          for (int w = 0; w < 4; ++w)
            keep = keep || (cc != 255 && c < t0 + w && tq[w] != 255 && cc != tq[w] && !(t0 + w < n_hub));
        }
      }
      const unsigned long long bal = __ballot(keep);
      if (keep) glist[count + __popcll(bal & ((1ull << lane) - 1ull))] = ent;
      count += __popcll(bal);
    }
    if (lane == 0) s_nlist = count;
  }

  // target operands (B operands of the MFMA): resident in registers; the last half fragment's
  // augmented columns (-u1, -u2, AUG, AUG) of the candidate form become (AUG, AUG, -u1, -u2)
  half8 th[NK];
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) th[ks] = A.F[((int64_t)t * NK + ks) * 64 + lane];
  if (hf) {
    const half8 x = th[NK - 1];
    half8 y = x;
    y[4] = x[6]; y[5] = x[7]; y[6] = x[4]; y[7] = x[5];
    th[NK - 1] = y;
  }
  const float thj = __uint_as_float(A.tinfo[(int64_t)t * 64 + l32]);   // theta of my column's row
  const int rowj = (int)A.tinfo[(int64_t)t * 64 + 32 + l32];
  const int mychr = (int)A.tchr[t] & 31;
  const unsigned int posj = (unsigned int)(t * 32 + l32);
  
  // column-direction list of my row (exclusive launches: register counter, see below)
  uint2 *mine = A.sl + (int64_t)(rowj >= 0 ? rowj : 0) * A.cap2;
  // (device-scope accesses: the previous item of this quad may have run on another XCD)
  int cntr = (excl && rowj >= 0) ? __hip_atomic_load(&A.cnt[rowj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
  // Records (row, partner position, d~ bits): hits that may not touch a row's counter directly
  // are staged per wave in LDS and flushed to a global pool with ONE returning atomic per STG
  // records; k_sym_regroup distributes them after the sweep.
  uint4 *stg_all = reinterpret_cast<uint4 *>(glist + A.glist_cap);
  uint4 *stg = stg_all + wave * STG;
  int scnt = 0;                                      // wave-uniform
  auto flush = [&]() {
    if (scnt > 0) {
      unsigned int base = 0;
      if (lane == 0) base = atomicAdd(A.pool_head, (unsigned int)scnt);
      base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
      for (int i = lane; i < scnt; i += 64)
        if (base + (unsigned int)i < A.pool_cap) A.pool[base + (unsigned int)i] = stg[i];
      if (base + (unsigned int)scnt > A.pool_cap && lane == 0) *A.pool_ovf = 1u;
      scnt = 0;
      __builtin_amdgcn_s_waitcnt(0x0F70);          // (nothing may look pending at the top of the loop)
    }
  };
  // reserve room for `total` records of one event: returns the staging offset of the event's
  // first record, or -(pool offset) - 1 when the event goes straight to the pool (total > STG)
  int direct_base = 0;
  auto reserve = [&](int total) -> int {
    if (total > STG) {
      unsigned int base = 0;
      if (lane == 0) base = atomicAdd(A.pool_head, (unsigned int)total);
      direct_base = __builtin_amdgcn_readfirstlane((int)base);
      if ((unsigned int)direct_base + (unsigned int)total > A.pool_cap && lane == 0) *A.pool_ovf = 1u;
      return 0;
    }
    if (scnt + total > STG) flush();
    const int o = scnt;
    scnt += total;
    return o;
  };
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // staging: stage_group (screen_common.h) + one tile-info piece per wave; every wave issues NPW loads
  static_assert(NPW == (CTG * NK + WPB - 1) / WPB + 1, "pieces per wave");
  const int piece0 = stage_piece0<CTG * NK, WPB>(wave_u);
  const int ip = CTG > 1 ? wave_u % CTG : 0;
  auto fetch = [&](int g, int slot) {
    stage_group<CTG * NK, WPB>(A.F + (int64_t)g * TILE_H8, sbuf + slot * TILE_H8, piece0, lane);
    __builtin_amdgcn_global_load_lds(A.tinfo + ((int64_t)g * CTG + ip) * 64 + lane,
                                     (__attribute__((address_space(3))) void *)(tinf + (slot * CTG + ip) * 64), 4, 0, 0);
  };
  __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): the target loads are history
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) asm volatile("" : "+v"(th[ks]));
#endif
  __syncthreads();
  const int n_my = s_nlist;
#pragma unroll
  for (int q = 0; q < RING - 1; ++q)
    if (q < n_my) fetch(glist[q] & 0xfffff, q);

  for (int q = 0; q < n_my; ++q) {
    const int cur = __builtin_amdgcn_readfirstlane(glist[q]);
    const int g = cur & 0xfffff;                     // (scalar: the tmin loads below must be SMEM --
    const int slot = q % RING;                       //  a vector load here would drain the prefetch)
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
      const int c = g * CTG + s;
      act[s] = c < t && ((cur >> (20 + 5 * s)) & 31) != mychr && !(t < n_hub);
      any_act = any_act || act[s];
    }
    if (!any_act) continue;                          // wave-uniform
    float thc[CTG];                                  // min theta over the streamed tile's rows
#pragma unroll
    for (int s = 0; s < CTG; ++s) thc[s] = A.tmin[g * CTG + s];

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
    if (A.dbg & 2) {   // (diagnostics: matrix pipe + staging only)
#pragma unroll
      for (int s = 0; s < CTG; ++s) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" ::"v"(acc[s]));
#endif
      }
      continue;
    }
#pragma unroll
    for (int s = 0; s < CTG; ++s) {
      if (!act[s]) continue;                         // wave-uniform
      // gate: can anything in this tile pair pass in either direction?
      float m = fmaxf(fmaxf(acc[s][0], acc[s][1]), acc[s][2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) m = fmaxf(fmaxf(m, acc[s][r]), acc[s][r + 1]);
      m = fmaxf(m, acc[s][15]);
      const bool col_gate = (g * CTG + s >= n_hub) && __any(m >= thj), row_gate = __any(m >= thc[s]);
      if (!(col_gate || row_gate) || (A.dbg & 1)) continue;
      ++C.n_slow;
      C.n_cg += col_gate ? 1u : 0u;
      C.n_rg += row_gate ? 1u : 0u;
      if (excl && col_gate && !row_gate) {
        // The common event -- column hits only, in an exclusive item (85 % of all events at 15 kb x 500):
        // one compare + one add per output for the lane's hit count (the gate is exact in this direction:
        // some lane has a hit), slots from the register counter, then per output one compare whose VCC
        // is both the scalar "anybody?" test and the store's lane mask -- no pass-bit words, no wave-wide
        // OR (six DPP steps), no second direction.  The event path costs the sweep 3.4 ms of its 21 (the
        // stores themselves 0.6: ablations), so its INSTRUCTION COUNT is what counts: the store loop
        // carries no capacity test (a row whose list would overflow compares against +inf instead of its
        // threshold -- it stores nothing and is flagged below, its list is never read) and walks a
        // pointer (one 64-bit add per hit output instead of sign extension + shift-add).
        unsigned int pc = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) pc += (acc[s][r] >= thj) ? 1u : 0u;
        ++C.n_ce;
        C.n_app += pc;
        const auto pcs = __builtin_amdgcn_permlane32_swap(pc, pc, false, false);
        uint2 *p = mine + (cntr + (hf ? (int)pcs[0] : 0));
        cntr += (int)(pcs[0] + pcs[1]);
        const float the = cntr <= A.cap2 ? thj : HUGE_VALF;
        const unsigned int cposb = (unsigned int)((g * CTG + s) * 32 + 4 * hf);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool hit = acc[s][r] >= the;
          if (__any(hit)) {
            asm volatile("" ::: "memory");               // (keeps the two tests separate)
            if (hit) {
              *p = make_uint2(__float_as_uint(-2.f * acc[s][r]), cposb + (unsigned int)(8 * (r >> 2) + (r & 3)));
              ++p;
            }
          }
        }
        continue;
      }
      const unsigned int *ti = tinf + (slot * CTG + s) * 64;
      // per-lane pass bits, bit (15 - r) = output r: column direction (my row is the target, the
      // streamed rows are candidates) and row direction (a streamed row is the target)
      unsigned int pm = 0, rm = 0;
      if (col_gate) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          pm = __builtin_amdgcn_alignbit(pm, ~__float_as_uint(acc[s][r] - thj), 31);
        pm &= 0xffffu;
      }
      if (row_gate) {
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4) {
          const uint4 tv = *reinterpret_cast<const uint4 *>(ti + 8 * a4 + 4 * hf);
          rm = __builtin_amdgcn_alignbit(rm, ~__float_as_uint(acc[s][4 * a4 + 0] - __uint_as_float(tv.x)), 31);
          rm = __builtin_amdgcn_alignbit(rm, ~__float_as_uint(acc[s][4 * a4 + 1] - __uint_as_float(tv.y)), 31);
          rm = __builtin_amdgcn_alignbit(rm, ~__float_as_uint(acc[s][4 * a4 + 2] - __uint_as_float(tv.z)), 31);
          rm = __builtin_amdgcn_alignbit(rm, ~__float_as_uint(acc[s][4 * a4 + 3] - __uint_as_float(tv.w)), 31);
        }
        rm &= 0xffffu;
      }
      const unsigned int both = wcx::wave_or_u32(pm | (rm << 16));
      if (both == 0) continue;
      const unsigned int anym = both & 0xffffu, rany = both >> 16;
      C.n_ce += anym ? 1u : 0u;
      C.n_re += rany ? 1u : 0u;
      const unsigned int pc = (unsigned int)__popc(pm), rc = (unsigned int)__popc(rm);
      C.n_app += pc + rc;
      C.n_row += rc;
      // destinations.  Exclusive launches: column hits go straight to my row's list under a register
      // counter (the target's two lanes l, l + 32 swap their pass counts); everything else becomes
      // records -- one reservation per event, column records first.
      int ofs = 0, coff = 0, roff = 0, rtotal = 0;
      if (excl) {
        if (anym) {
          const auto pcs = __builtin_amdgcn_permlane32_swap(pc, pc, false, false);
          ofs = cntr + (hf ? (int)pcs[0] : 0);
          cntr += (int)(pcs[0] + pcs[1]);
          if (cntr > A.cap2) pm = 0;               // (overflowing row: flagged below, nothing stored)
        }
        if (rany) {
          const int incl = wcx::wave_incl_scan_i((int)rc);
          rtotal = __builtin_amdgcn_readlane(incl, 63);
          roff = reserve(rtotal) + incl - (int)rc;
        }
      } else {
        const int incl = wcx::wave_incl_scan_i((int)(pc | (rc << 16)));   // both sums at once (< 2^16 each)
        const int tot = __builtin_amdgcn_readlane(incl, 63);
        const int tc = tot & 0xffff, tr = tot >> 16;
        rtotal = tc + tr;
        const int b0 = reserve(rtotal);
        coff = b0 + (incl & 0xffff) - (int)pc;
        roff = b0 + tc + (incl >> 16) - (int)rc;
      }
      const unsigned int cposb = (unsigned int)((g * CTG + s) * 32 + 4 * hf);
      // One scalar-skipped, fully unrolled pass over the 16 outputs (static register indices; a
      // per-lane walk of the hits through LDS was measured slower: its dependent LDS reads are
      // exposed, 2 000 SIMD cycles per event at K = 512).
      // records of this event go to the staging area (LDS) or, for an event bigger than it, straight
      // to the pool: two separate loops, so that every store has ONE address space
      const bool direct = rtotal > STG;
      uint4 *rec_dst = direct ? A.pool + (unsigned int)direct_base : nullptr;
      const unsigned int rec_room = direct ? (A.pool_cap > (unsigned int)direct_base ? A.pool_cap - (unsigned int)direct_base : 0u) : 0u;
      if (anym) {
        if (excl) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (anym & (0x8000u >> r)) {
              asm volatile("" ::: "memory");               // (keeps the two tests separate)
              if (pm & (0x8000u >> r)) {
                mine[ofs] = make_uint2(__float_as_uint(-2.f * acc[s][r]),
                                       cposb + (unsigned int)(8 * (r >> 2) + (r & 3)));
                ++ofs;
              }
            }
          }
        } else if (!direct) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (anym & (0x8000u >> r)) {
              asm volatile("" ::: "memory");
              if (pm & (0x8000u >> r)) {
                stg[coff] = make_uint4((unsigned int)rowj, cposb + (unsigned int)(8 * (r >> 2) + (r & 3)),
                                       __float_as_uint(-2.f * acc[s][r]), 0u);
                ++coff;
              }
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (anym & (0x8000u >> r)) {
              asm volatile("" ::: "memory");
              if (pm & (0x8000u >> r)) {
                if ((unsigned int)coff < rec_room)
                  rec_dst[coff] = make_uint4((unsigned int)rowj, cposb + (unsigned int)(8 * (r >> 2) + (r & 3)),
                                             __float_as_uint(-2.f * acc[s][r]), 0u);
                ++coff;
              }
            }
          }
        }
      }
      if (rany) {
        if (!direct) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (rany & (0x8000u >> r)) {
              asm volatile("" ::: "memory");
              if (rm & (0x8000u >> r)) {
                stg[roff] = make_uint4(ti[32 + 8 * (r >> 2) + 4 * hf + (r & 3)], posj,
                                       __float_as_uint(-2.f * acc[s][r]), 0u);
                ++roff;
              }
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (rany & (0x8000u >> r)) {
              asm volatile("" ::: "memory");
              if (rm & (0x8000u >> r)) {
                if ((unsigned int)roff < rec_room)
                  rec_dst[roff] = make_uint4(ti[32 + 8 * (r >> 2) + 4 * hf + (r & 3)], posj,
                                             __float_as_uint(-2.f * acc[s][r]), 0u);
                ++roff;
              }
            }
          }
        }
      }
    }
  }
  flush();
  if (excl && hf == 0 && rowj >= 0) {
    __hip_atomic_store(&A.cnt[rowj], cntr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cntr > A.cap2) A.flags[rowj] = 1u;
  }
}

// Persistent kernel: the workgroups pull work items (chunk-major: all quads above chunk 0, then
// chunk 1, ...) from a device-side queue, so the sweep has no launch boundaries -- no partly filled
// last rounds -- while the workgroups running at any time still stream the same one or two chunks
// (L2 sharing).  Exclusive items of one quad (its chunks 0, 1, 2, ...) are ordered by a per-quad
// sequence counter: item (quad, l) starts after (quad, l - 1) has published its rows' counters
// (write-through stores, vmcnt(0), then the counter: device-scope on both sides).  A waiting item
// only ever waits for an item that was dequeued before it, so the scheme cannot deadlock.
// LBW = waves per SIMD budgeted.
template <int NK, int CTG, int LBW, int RING>
__global__ __launch_bounds__(256, LBW) void k_screen_sym(const SymArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int s_nlist, s_item, s_desc;
  const int tid = threadIdx.x;
  if (A.gate && !*A.gate) return;                      // (a second attempt nobody asked for)
  const int n_tiles = (int)A.glob->n_tiles;
  SymCounters C = {0, 0, 0, 0, 0, 0, 0};
  for (;;) {
    __syncthreads();                                   // everybody is done with the previous item
    if (tid == 0) s_item = (int)atomicAdd(A.queue_head, 1u);
    __syncthreads();
    const int item_g = s_item;
    if (item_g >= A.total_items) break;
    if (A.n_parts > 1 && item_g % A.n_parts != A.part) continue;      // another rank's work item
    if (tid < A.n_desc && item_g >= A.desc[tid].item_base &&
        item_g < A.desc[tid].item_base + A.desc[tid].n_q * A.desc[tid].n_split)
      s_desc = tid;
    __syncthreads();
    const SymDesc d = A.desc[s_desc];
    const int item = item_g - d.item_base;
    const int quad = d.q_first + item / d.n_split, split = item % d.n_split;
    if (quad * 4 >= n_tiles) continue;                 // (the tables are sized for a tile bound)
    const int excl = (d.n_split == 1 && !A.force_records) ? 1 : 0;
    if (excl) {
      if (tid == 0) {
        while (__hip_atomic_load(&A.seq[quad], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != d.index)
          __builtin_amdgcn_s_sleep(16);
      }
      __syncthreads();
    }
    sym_item<NK, CTG, RING>(A, smem, &s_nlist, quad, split, d.n_split, d.c0, d.c1, excl, C);
    if (excl) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's counter stores have left
      __syncthreads();
      if (tid == 0) __hip_atomic_store(&A.seq[quad], d.index + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (A.stats) {
    const int lane = tid & 63;
    const int tot_a = wcx::wave_sum_i((int)C.n_app), tot_r = wcx::wave_sum_i((int)C.n_row);
    if (lane == 0) {
      atomicAdd(&A.stats[4], (unsigned long long)tot_a);
      atomicAdd(&A.stats[7], (unsigned long long)tot_r);
      atomicAdd(&A.stats[6], (unsigned long long)C.n_slow);
      atomicAdd(&A.stats[8], (unsigned long long)C.n_cg);
      atomicAdd(&A.stats[9], (unsigned long long)C.n_rg);
      atomicAdd(&A.stats[10], (unsigned long long)C.n_ce);
      atomicAdd(&A.stats[11], (unsigned long long)C.n_re);
    }
  }
}

template <int NK, int CTG, int LBW, int RING>
int sym_launch_t(const SymArgs &a, unsigned grid, size_t lds, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_screen_sym<NK, CTG, LBW, RING>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  k_screen_sym<NK, CTG, LBW, RING><<<grid, 256, lds, st>>>(a);
  return (int)hipGetLastError();
}
#define WCX_SYM_TRY(N, C, L, R) \
  if (nk == N && ctg == C && lb == L && ring == R) return sym_launch_t<N, C, L, R>(a, grid, lds, st);

}  // namespace
