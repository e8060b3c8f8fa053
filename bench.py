#!/usr/bin/env python3
"""bench.py -- the WisecondorX newref+predict hot path on MI355X.

Contract (see the task prompt): `python bench.py --gpus N --steps K --warmup W` prints ONE
JSON line on rank 0.  A "step" = one pass of the hot path over one synthetic batch = the whole
`newref` + one `predict`:
  newref   the three passes of main.py:82-130 -- A (all samples, every autosomal bin), F (the female
           samples, chrX rows), M (the male samples, chrX + chrY rows): reference-bin search
           (all-pairs distance + top-k, k=300) + null-ratio table per pass.  Target rows split over
           the N ranks (A: the reference's own _get_part formula, newref_tools.py:244-247; F / M:
           their gonosomal rows) after ONE RCCL all-gather per pass of the row-sharded bin-feature
           matrix X; the finished row blocks are all-gathered so that every rank holds the whole
           reference (what `newref` writes to disk);
  predict  one (female) test sample against that reference, complete and device-resident: cut-off,
           weights, three masked normalisation passes for the autosomes and for the gonosomes, the
           A + gonosome merge, post-processing, CBS segmentation of 23 chromosomes, segment
           z-scores (replicated on every rank: predict has no collective on its path).
Default workload = north_star's headline problem, the size of BASELINE.json configs[3]:
15 kb bins (hg38, ~5 % of bins masked) x 500 reference samples, refsize 300 -- it fits one
GPU; with --gpus N the same problem is row-sharded (strong scaling).  Inputs are resident in
HBM when the timed region starts.  value = candidate bin pairs evaluated per second over the
whole job ("bins x refs / s").  A secondary block carries configs[2] (100 samples).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # MI355X datasheet FP64 vector == matrix (not in the guide)
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense BF16/FP16 MFMA
SPINUP_STEPS = int(os.environ.get("WCX_BENCH_SPINUP_STEPS", "0"))   # extra untimed steps (0: only --warmup)
TRAFFIC_JSONS = [os.path.join(ROOT, "profiles", r_, "screen_traffic.json") for r_ in ("r06", "r05", "r04")]
SIMD_CLOCK_HZ = 2.4e9          # MI355X_MICROARCH.md: 2.4 GHz peak engine clock (1 024 SIMDs)


def screen_source_sha():
    """Hash of the screen kernel's sources: roofline.traffic (PMC counters recorded by
    scripts/measure_traffic.sh) is only reported while the kernel is the one that was profiled."""
    h = hashlib.sha256()
    for f in ("screen_kernel.h", "screen_sym.h", "screen_count.h", "screen_common.h", "newref_topk_screen.hip"):
        h.update(open(os.path.join(ROOT, "wisecondorx_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def make_workload(binsize, n_samples, seed=0, device=0):
    """Synthetic cohort -> masked, depth-normalised, PCA-corrected X of the autosomal pass (host,
    untimed).  (The single-pass form: scripts/ and tests use it.)"""
    from wisecondorx_amd import prep
    from wisecondorx_amd.synth import Cohort
    co = Cohort(binsize, struct_seed=1234 + seed)
    samples, genders = co.cohort(n_samples, seed0=100 + seed)
    mask, bpc = prep.get_mask(samples)
    from wisecondorx_amd import _lib
    # PCA stage on THIS rank's GPU (the library selects its context's device on every call, and
    # torch follows the runtime's current device: a rank must never touch another rank's device)
    p = prep.prepare(samples, "A", mask, bpc, ctx=_lib.default_context(device))
    test = co.sample(777 + seed, "F", cnv=[(3, 100, 100 + max(4, int(4e7 // binsize)), 1.5)])
    return co, p, test


def make_full_workload(binsize, n_samples, seed=0, device=0):
    """The inputs of the WHOLE newref (main.py:82-130): gender-corrected cohort, the shared mask,
    and the prepared (masked, depth-normalised, PCA-corrected) matrices of the A, F and M passes;
    + one female test sample with a planted gain.  Host + PCA on this rank's GPU, untimed."""
    from wisecondorx_amd import _lib, prep
    from wisecondorx_amd.overall_tools import gender_correct
    from wisecondorx_amd.synth import Cohort
    # (female chrY coverage 10 % of normal: with the default 0.2 % every chrY bin fails the
    # female-only mask and the M pass has no chrY target rows)
    co = Cohort(binsize, struct_seed=1234 + seed, female_y=0.1)
    samples, genders = co.cohort(n_samples, seed0=100 + seed)
    samples = np.array([gender_correct(s_, g_) for s_, g_ in zip(samples, genders)])
    g = np.array(genders)
    total_mask, bpc = prep.get_mask(samples)
    total_mask = total_mask & prep.get_mask(samples[g == "F"])[0] & prep.get_mask(samples[g == "M"])[0]
    ctx = _lib.default_context(device)
    n_aut = int(np.sum(bpc[:22]))
    passes = {"A": prep.prepare(samples, "A", total_mask, bpc, ctx=ctx)}
    passes["F"] = prep.prepare(samples[g == "F"], "F", total_mask, bpc, ctx=ctx, frozen=n_aut)
    passes["M"] = prep.prepare(samples[g == "M"], "M", total_mask, bpc, ctx=ctx, frozen=n_aut)
    test = gender_correct(co.sample(777 + seed, "F", cnv=[(3, 100, 100 + max(4, int(4e7 // binsize)), 1.5)]), "F")
    co.cohort_corrected = (samples, genders)       # (for the CLI end-to-end block of bench.py)
    return co, passes, test


def cpu_baseline(Xs, chr_cum, k, budget_s=10.0):
    """The reference's per-bin search on the host cores, single process (the reference is
    single-threaded in effect: its --cpus threads are GIL-bound), on a bounded
    sample of target rows, each against ALL its candidate rows:
      value   the NumPy + Python-scan restatement (oracle.wcx_oracle.sq_distances + topk_scan:
              what newref_tools.py:260-275 executes), the primary figure;
      c_port  the plain-C port of the same loop (oracle/wcx_oracle.c)."""
    from oracle import c_oracle as CO
    from oracle import wcx_oracle as O
    B = Xs.shape[1]
    X = Xs.T                                      # (B, S) Fortran-ordered view
    rng = np.random.default_rng(0)
    rows = rng.choice(B, 4096, replace=False)

    def own(t):
        c = int(np.searchsorted(chr_cum, t, side="right"))
        return (int(chr_cum[c - 1]) if c else 0), int(chr_cum[c])
    t0 = time.perf_counter()
    pairs_py, n_py = 0, 0
    for t in rows:
        cs, ce = own(int(t))
        chr_data = np.concatenate((X[:cs], X[ce:]))          # newref_tools.py:192-199
        O.topk_scan(O.sq_distances(chr_data, X[int(t), :]), k)
        pairs_py += B - (ce - cs)
        n_py += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt_py = time.perf_counter() - t0
    t0 = time.perf_counter()
    pairs_c, n_c = 0, 0
    for t in rows:
        cs, ce = own(int(t))
        CO.topk_rows(Xs, cs, ce, int(t), int(t) + 1, k)
        pairs_c += B - (ce - cs)
        n_c += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt_c = time.perf_counter() - t0
    # the cache-tiled C port on every host thread the cgroup allows (what a tuned CPU build would do)
    nth = CO.host_threads()
    starts = np.sort(rng.choice(B - 32, 4 * nth, replace=False))
    t0 = time.perf_counter()
    rows_mt, _, _ = CO.topk_row_blocks_threaded(Xs, chr_cum, starts, 32, k, threads=nth)
    dt_mt = time.perf_counter() - t0
    pairs_mt = sum(B - (own(int(t))[1] - own(int(t))[0]) for t in rows_mt)
    mb = np.diff(np.concatenate(([0], chr_cum)))
    return {"value": pairs_py / dt_py, "unit": "bin-pairs/s", "cores": 1, "kind": "port",
            "host_cores_available": os.cpu_count(),
            "sample": "{} random target rows x all {} candidate rows, S={}, k={}: NumPy distance + "
                      "Python scan restatement of newref_tools.py:260-275 ({:.1f} s); "
                      "single-threaded like the reference".format(n_py, B, Xs.shape[0], k, dt_py),
            "c_port": {"value": pairs_c / dt_c, "rows": n_c, "seconds": dt_c,
                       "what": "oracle/wcx_oracle.c, the same loop in C, 1 core"},
            "c_port_tiled_mt": {"value": pairs_mt / dt_mt, "rows": int(len(rows_mt)), "seconds": dt_mt,
                                "threads": nth,
                                "what": "oracle/wcx_oracle_tiled.c (same arithmetic, cache-tiled) on the "
                                        "host threads the cgroup CPU quota allows"},
            "extrapolated_full_search_s": float(np.sum(mb * (B - mb))) / (pairs_py / dt_py)}


def verify_rows(w, n_blocks=128, rows_per_block=16, gon_blocks=40):
    """Out of the timed region: the reference-bin tables of the LAST timed step against the C oracle
    (oracle/wcx_oracle_tiled.c on the host's cores) -- indices and distances bit for bit, every row
    against all its candidates: n_blocks x rows_per_block scattered target rows of the A pass, and
    gon_blocks x rows_per_block scattered gonosomal target rows of the F pass (chrX) and of the M pass
    (chrX + chrY, the chrY rows all included in the draw's range)."""
    from oracle import c_oracle as CO
    if w.last is None:
        return None
    rng = np.random.default_rng(11)
    starts = np.sort(rng.choice(w.B - rows_per_block, n_blocks, replace=False))
    t0 = time.perf_counter()
    rows, oi, od = CO.topk_row_blocks_threaded(w.Xs_host, w.cum, starts, rows_per_block, w.k)
    sel = w.torch.from_numpy(rows).to(w.last[0].device)
    gi = w.last[0][sel].cpu().numpy()
    gd = w.last[1][sel].cpu().numpy()
    bad = int(np.count_nonzero((gi != oi).any(axis=1) | (gd != od).any(axis=1)))
    out = {"rows": int(len(rows)), "mismatches": bad}
    n_all, bad_all = int(len(rows)), bad
    ref = getattr(w, "last_ref", None) or {}
    if w.world == 1:
        for tag in ("F", "M"):
            if tag not in ref:
                continue
            P = w.P[tag]
            cum, Bp = P["cum"], int(P["B"])
            g0 = int(cum[21])                                  # first gonosomal row of this pass
            if Bp - g0 < rows_per_block:
                continue
            Xs = np.ascontiguousarray(P["p"]["X"].T)           # [S][B] of this pass
            n_b = min(gon_blocks, (Bp - g0) // rows_per_block)
            st = g0 + np.sort(rng.choice(Bp - g0 - rows_per_block + 1, n_b, replace=False))
            if tag == "M" and len(cum) > 23 and Bp - int(cum[22]) >= rows_per_block:
                st[-1] = Bp - rows_per_block                   # the last chrY rows are always in
                st = np.unique(st)
            r_, oi, od = CO.topk_row_blocks_threaded(Xs, cum, st, rows_per_block, w.k)
            sel = w.torch.from_numpy(r_).to(ref[tag]["idx"].device)
            gi = ref[tag]["idx"][sel].cpu().numpy()
            gd = ref[tag]["dist"][sel].cpu().numpy()
            b_ = int(np.count_nonzero((gi != oi).any(axis=1) | (gd != od).any(axis=1)))
            out[tag + "_pass"] = {"rows": int(len(r_)), "mismatches": b_,
                                  "chrY_rows": int(np.count_nonzero(r_ >= int(cum[22]))) if len(cum) > 23 else 0}
            n_all += int(len(r_))
            bad_all += b_
    out.update({"rows_all_passes": n_all, "mismatches_all_passes": bad_all,
                "seconds": time.perf_counter() - t0,
                "what": "indices and distances of {} target rows of the A pass ({} scattered blocks) and of "
                        "scattered gonosomal rows of the F and M passes of the last timed step, bit for bit "
                        "against oracle/wcx_oracle_tiled.c on {} host threads (each row against all its "
                        "candidates)".format(len(rows), n_blocks, CO.host_threads())})
    return out


class Workload:
    """Device-resident inputs and the step of one problem size."""

    def __init__(self, args, n_samples, torch, dev, dev_index, rank, world):
        from wisecondorx_amd import _lib, predict_tools
        from wisecondorx_amd import dist as wd
        from wisecondorx_amd.newref_tools import _get_part
        self.torch, self.wd, self.pt = torch, wd, predict_tools
        self.rank, self.world, self.args = rank, world, args
        co, passes, test = make_full_workload(args.binsize, n_samples, device=dev_index)
        self.co = co
        self.k = args.refsize
        stream = torch.cuda.current_stream().cuda_stream
        self.ctx = _lib.Context(dev_index, stream)
        if args.debug_flags:
            self.ctx.lib.wcx_debug_flags(self.ctx.h, args.debug_flags)
        self.backend = wd.GpuBackend(self.ctx)
        # the gonosomal passes on their own contexts / streams, side by side with the autosomal pass
        # (they search ~10 k rows each: less than one round of workgroups)
        self.side = {}
        if world == 1 and args.concurrent_passes and not args.debug_flags:
            for tag in ("F", "M"):
                st_ = torch.cuda.Stream(device=dev)
                cx_ = _lib.Context(dev_index, st_.cuda_stream)
                self.side[tag] = (st_, cx_, wd.GpuBackend(cx_))
            self.sweep_event = _lib.vp()
            _lib.check(self.ctx.lib.wcx_sweep_event(self.ctx.h, _lib.C.byref(self.sweep_event)))
        self.P = {}
        self.pairs_total = 0
        for tag in ("A", "F", "M"):
            p = passes[tag]
            X = p["X"]                                   # (B, S) Fortran order
            B, S = X.shape
            cum = np.asarray(p["masked_bins_per_chr_cum"], dtype=np.int64)
            mb = np.asarray(p["masked_bins_per_chr"], dtype=np.int64)
            tgt = mb if tag == "A" else mb * (np.arange(len(mb)) >= 22)     # searched chromosomes
            pairs = int(np.sum(tgt * (B - mb)))
            self.pairs_total += pairs
            rb, re_ = _get_part(rank, world, B)
            shard_rows = wd.max_shard_rows(world, B)
            Xrow = torch.zeros((shard_rows, S), dtype=torch.float64, device=dev)
            Xrow[:re_ - rb] = torch.from_numpy(np.ascontiguousarray(X[rb:re_])).to(dev)
            m = min(S, 100)
            ids = np.ascontiguousarray(np.random.default_rng(5).permutation(S)[:m], dtype=np.int32)
            full_rows = shard_rows if tag == "A" else B
            self.P[tag] = {"p": p, "B": B, "S": S, "cum": cum, "pairs": pairs, "Xrow": Xrow, "ids": ids,
                           "bufs": (torch.empty((full_rows, self.k), dtype=torch.int32, device=dev),
                                    torch.empty((full_rows, self.k), dtype=torch.float64, device=dev),
                                    torch.empty((full_rows, m), dtype=torch.float64, device=dev))}
        pa = passes["A"]
        self.p = pa
        self.Xs_host = np.ascontiguousarray(pa["X"].T)   # [S][B] of the autosomal pass (verify, cpu_baseline)
        self.S, self.B = self.Xs_host.shape
        self.cum = self.P["A"]["cum"]
        ref = dict(pa)
        ref.update({k_ + ".F": v for k_, v in passes["F"].items()})
        pt = predict_tools
        xA = pt.project_pc(pt.coverage_normalize_and_mask(test, ref, ""), ref, "")
        xG = pt.project_pc(pt.coverage_normalize_and_mask(test, ref, ".F"), ref, ".F")
        self.d_xA = torch.from_numpy(np.ascontiguousarray(xA)).to(dev)
        self.d_xG = torch.from_numpy(np.ascontiguousarray(xG)).to(dev)
        self.pargs = argparse.Namespace(minrefbins=150, alpha=1e-4, seed=1, maskrepeats=5)
        self.rem = {"args": self.pargs, "mask": passes["F"]["mask"], "bins_per_chr": passes["F"]["bins_per_chr"],
                    "binsize": args.binsize, "ref_gender": "F"}
        self.names = ("topk", "topk_screen", "topk_pre", "topk_prep", "topk_refine", "null_ratios")
        self.ms = {"{}:{}".format(t, n_): [] for t in ("A", "F", "M") for n_ in self.names}
        for n_ in ("normalize", "cbs", "segment_z", "predict_full", "gather_ref", "cutoff", "weights"):
            self.ms[n_] = []
        self.fb_rows = []
        self.n_segments = 0
        self.last = None                             # (idx, dist) of the last step's A pass, for verify_rows
        self.stats_A = None

    def step(self, record):
        from wisecondorx_amd import _lib as _lib_
        wd, ctx, torch = self.wd, self.ctx, self.torch
        ref = {}
        side, done = self.side, {}
        for tag in ("A", "F", "M"):
            P = self.P[tag]
            if tag in side:
                # starts when the A pass's sweep is done: its MFMA sweep runs beside A's L2-bound refine
                st_, cx_, be_ = side[tag]
                cx_.timer_tag(tag + ":")
                with torch.cuda.stream(st_):
                    _lib_.check(ctx.lib.wcx_wait_event(cx_.h, self.sweep_event))
                    idx, dist_, nr = wd.newref_gonosomal_sharded(P["Xrow"], P["B"], P["cum"], self.k, P["ids"],
                                                                 be_, self.rank, self.world, P["bufs"])
                    done[tag] = torch.cuda.Event()
                    done[tag].record()
                cx_.timer_tag("")
                ref[tag] = {"idx": idx, "dist": dist_, "nr": nr, "cum": P["cum"]}
                continue
            ctx.timer_tag(tag + ":")
            if tag == "A":
                # ONE exchange (all-gather of the row shards of X over RCCL/xGMI), the search + null
                # ratios of this rank's target rows, then every rank gets the whole tables
                # (N > 1: the tile pairs of the symmetric sweep dealt out to the ranks + ONE all-to-all of
                #  the hit records, dist.newref_sym_sharded; WCX_BENCH_SYM_SHARD=0: every rank sweeps its
                #  target rows against all candidates, the round-1..4 form)
                shard_fn = wd.newref_sym_sharded if ((self.world > 1 or wd.force_collectives()) and
                                                     os.environ.get("WCX_BENCH_SYM_SHARD", "1") != "0") \
                    else wd.newref_sharded
                idx_l, dist_l, nr_l, _ = shard_fn(P["Xrow"], P["B"], P["cum"], self.k, P["ids"],
                                                  self.backend, self.rank, self.world, out=P["bufs"])
                if self.args.debug_flags & 123:           # (ablations leave garbage neighbour tables)
                    ctx.timer_tag("")
                    ctx.sync()
                    if record:
                        self.ms["A:topk"].append(ctx.kernel_ms("A:topk"))
                        self.ms["A:topk_screen"].append(ctx.kernel_ms("A:topk_screen"))
                        self.ms["A:topk_pre"].append(ctx.kernel_ms("A:topk_pre"))
                    return
                if record and self.world > 1:
                    torch.cuda.synchronize()             # (the gather's own time, not the drain of the search)
                t0 = time.perf_counter()
                idx, dist_, nr = wd.gather_reference3(idx_l, dist_l, nr_l, P["B"], self.world, self.backend)
                if record and self.world > 1:
                    torch.cuda.synchronize()
                    self.ms["gather_ref"].append(1e3 * (time.perf_counter() - t0))
                self.last = (idx, dist_)
                if record and self.stats_A is None:
                    # counters of the A pass (rows, appends, fallbacks): read once, in the first timed
                    # step, before the F pass resets them (this read synchronises)
                    self.stats_A = ctx.topk_stats()
            else:
                idx, dist_, nr = wd.newref_gonosomal_sharded(P["Xrow"], P["B"], P["cum"], self.k, P["ids"],
                                                             self.backend, self.rank, self.world, P["bufs"])
            ref[tag] = {"idx": idx, "dist": dist_, "nr": nr, "cum": P["cum"]}
        ctx.timer_tag("")
        for tag in done:
            torch.cuda.current_stream().wait_event(done[tag])
        self.last_ref = ref
        # predict ONE sample, complete (replica on every rank): autosomes vs A, gonosomes vs F
        t0 = time.perf_counter()
        res = wd.predict_full_dev(self.backend, ref["A"], ref["F"], self.d_xA, self.d_xG, self.rem, self.pt)
        self.n_segments = len(res)
        if record:
            self.ms["predict_full"].append(1e3 * (time.perf_counter() - t0))
            self.ms["normalize"].append(ctx.kernel_ms("aut:normalize") + max(0.0, ctx.kernel_ms("normalize")))
            for name in ("cbs", "segment_z", "cutoff"):
                self.ms[name].append(ctx.kernel_ms(name))
            # two calls (autosomal + gonosomal reference), each with its own timer
            self.ms["weights"].append(ctx.kernel_ms("weights") + max(0.0, ctx.kernel_ms("gon:weights")))
            for tag in ("A", "F", "M"):
                cx_ = self.side[tag][1] if tag in self.side else ctx
                for n_ in self.names:
                    self.ms["{}:{}".format(tag, n_)].append(cx_.kernel_ms("{}:{}".format(tag, n_)))
            self.fb_rows.append(self.stats_A["fallback_rows"] if self.stats_A else -1)

    def mean_ms(self, name):
        v = [x for x in self.ms.get(name, []) if x is not None and x >= 0]
        return float(np.mean(v)) if v else -1.0

    def roofline(self):
        S = self.S
        stats = self.stats_A or self.ctx.topk_stats()
        k_ms = self.mean_ms("A:topk")
        screen_ms = self.mean_ms("A:topk_screen")
        pairs_A = self.P["A"]["pairs"]
        if self.world > 1:
            pairs_A = stats["pairs"]                     # this rank's share
        if screen_ms >= 0:
            # MFMA screen: algorithmic work = the -2 X^T X Gram GEMM, 2*S flop per candidate pair
            # (SURVEY.md §8d).  The kernel executes one fp16 product over K = 16*NK >= S + 4 (four
            # augmented columns carry the norm and the threshold), reported as executed_tflops.
            nk_list = [1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64]
            nk = next(v for v in nk_list if 16 * v >= S + 4)   # same rule as wcx_topk_screen_launch
            flops = 2.0 * S * pairs_A
            achieved = flops / (screen_ms * 1e-3) / 1e12
            sym = stats.get("sym_gates", 0) > 0
            r = {"kernel": ("k_screen_count (thresholds from counts over the low-norm rows) + k_screen_sym (symmetric "
                            "sweep: every tile pair once, both directions) + k_sym_regroup + k_sym_final" if sym
                            else "k_screen_hub1 (thresholds from counts over the low-norm rows) + k_screen") +
                           " of the autosomal pass (v_mfma_f32_32x32x16_f16, fp16 hi plane, fp32 acc) + fused "
                           "top-k filter", "bound": "mfma", "achieved": achieved,
                 "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": achieved / F16_MFMA_PEAK_TFLOPS, "traffic": None, "kernel_ms": screen_ms,
                 "executed_tflops": (0.5 + 1.0 / 16 if sym else 1.0) * 2.0 * nk * 16 * pairs_A / (screen_ms * 1e-3) / 1e12,
                 "prep_ms": self.mean_ms("A:topk_prep"), "refine_ms": self.mean_ms("A:topk_refine"),
                 "topk_total_ms": k_ms, "pairs_per_launch": pairs_A,
                 "concurrent_passes_in_timed_steps": bool(self.side),
                 "fallback_rows": stats["fallback_rows"], "fallback_rows_per_step": self.fb_rows,
                 "compactions": stats["compactions"], "appends": stats["appends"],
                 "refined_pairs": stats["refined"], "sym_gates": stats.get("sym_gates", 0),
                 "sym_row_appends": stats.get("sym_row_appends", 0), "sym_counts": stats["phase_cycles"][:4], "pre_ms": self.mean_ms("A:topk_pre"),
                 "kernel_ms_note": "wall time of one whole sweep of the A pass (HIP events on the launch "
                                   "stream): symmetric path = k_screen_count + ONE persistent k_screen_sym launch "
                                   "+ k_sym_regroup + k_sym_final (rocprofv3: the sum of those dispatches); "
                                   "one-directional path = k_screen_hub1 + chunk launches on two concurrent "
                                   "streams (average launch duration x launches per sweep / 2)",
                 "attainable_note": "the dense-f16 peak is not attainable on random data: a bare "
                                    "LDS-fed MFMA loop with DMA staging and no epilogue is power-limited "
                                    "to 1.1-1.35 PFLOP/s on this chip, box to box (1.9-2.0 on all-zero "
                                    "operands; profiles/r02/ubench_mfma.txt); the vendor's plain fp16 GEMM "
                                    "(torch.matmul -> hipBLASLt) sustains 1.24-1.31 PFLOP/s on random data on "
                                    "the same boxes (profiles/r03/gemm_calibration.json)"}
            if any(stats["phase_cycles"]):          # only with --debug-flags 4
                r["phase_cycles"] = stats["phase_cycles"]
        else:
            flops = 3.0 * S * pairs_A                # exact path: sub, mul, add per (pair, sample)
            achieved = flops / (k_ms * 1e-3) / 1e12
            r = {"kernel": "k_topk_exact (fp64 VALU, 3 flop per pair-sample)", "bound": "mfma",
                 "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": achieved / FP64_PEAK_TFLOPS, "traffic": None, "kernel_ms": k_ms,
                 "pairs_per_launch": pairs_A, "compactions": stats["compactions"]}
        r["null_ratios_ms"] = self.mean_ms("A:null_ratios")
        for name in ("normalize", "cbs", "segment_z", "gather_ref"):
            if self.ms[name]:
                r[name + "_ms"] = self.mean_ms(name)
        if self.ms["predict_full"]:
            # host wall-clock of the predict call; the newref kernels queued before it are still
            # running when it starts, so this is NOT the predict's own duration (that is ~6 ms:
            # profiles/r03 kernel trace)
            r["predict_call_wall_ms_incl_drain_of_newref"] = self.mean_ms("predict_full")
        r["gonosomal_passes"] = {
            tag: {"rows": int(self.P[tag]["B"] - int(self.P[tag]["cum"][21])), "samples": int(self.P[tag]["S"]),
                  "pairs": self.P[tag]["pairs"], "topk_ms": self.mean_ms(tag + ":topk"),
                  "screen_ms": self.mean_ms(tag + ":topk_screen"), "refine_ms": self.mean_ms(tag + ":topk_refine"),
                  "null_ratios_ms": self.mean_ms(tag + ":null_ratios")} for tag in ("F", "M")}
        return r, screen_ms



HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def hbm_kernels(w, rf):
    """the gather / scan kernels are HBM bound in principle -- achieved GB/s on their
    ALGORITHMIC bytes and the fraction of 8 TB/s (what they really move goes through L2: DESIGN.md 4)."""
    B, S, k = int(w.B), int(w.S), int(w.k)
    m = int(len(w.P["A"]["ids"]))
    BG = int(w.P["F"]["B"] - int(w.P["F"]["cum"][21]))
    rows = {
        "k_null_ratios (A pass)": (rf.get("null_ratios_ms", -1), B * k * 4 + m * B * 8,
                                   "B k 4 (indexes) + m B 8 (ratios out)"),
        "k_refine (A pass)": (rf.get("refine_ms", -1), B * S * 8 + int(rf.get("refined_pairs", 0)) * 8 + B * k * 12,
                              "X once + shortlists + idx / dist out (it gathers refined_pairs x S x 8 = "
                              "{:.0f} GB through L2)".format(int(rf.get("refined_pairs", 0)) * S * 8 / 1e9)),
        "k_cut_partial (cut-off, 5 repeats x 2 sweeps)": (w.mean_ms("cutoff"), 10 * B * k * 8,
                                                          "10 sweeps of the autosomal distances"),
        "k_weights": (w.mean_ms("weights"), (B + int(w.P["F"]["B"])) * k * 8,
                      "one sweep of distances: the autosomal + the gonosomal reference (two calls, both timed)"),
        "k_normalize_pass x3 (1 sample, A + gonosomes)": (w.mean_ms("normalize"), 3 * (B + BG) * (k * 4 + k // 8 + 32),
                                                          "per pass and bin: k indexes + selection bits + in / out values"),
    }
    out = {}
    for name, (ms, nbytes, what) in rows.items():
        if ms is None or ms <= 0:
            continue
        gbs = nbytes / (ms * 1e-3) / 1e9
        out[name] = {"ms": ms, "algorithmic_bytes": int(nbytes), "achieved_GBs": gbs,
                     "frac_of_hbm_peak": gbs / HBM_PEAK_GBS, "bytes": what}
    return out


def config5_block(w, batch=96, runs=3, verify=True):
    """BASELINE configs[4]: predict a batch of 96 samples at 15 kb, device-resident end to end
    (dist.predict_batch_dev) against the reference of the last timed step, on this one device (8 GPUs
    stripe the samples, no collective)."""
    import torch
    from wisecondorx_amd.overall_tools import gender_correct
    pt, wd = w.pt, w.wd
    A, G = w.last_ref["A"], w.last_ref["F"]
    ref = dict(w.p)
    ref.update({k_ + ".F": v for k_, v in w.P["F"]["p"].items()})
    tests = [gender_correct(w.co.sample(5000 + i, "F", cnv=[(1 + i % 22, 200, 2200, 1.5)]), "F")
             for i in range(batch)]
    dev = A["idx"].device
    cache = {}
    t_all, t_prep = [], []
    rows = None
    for _ in range(runs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dA, dG = pt.batch_counts_dev(tests, ref, ("", ".F"), dev, cache)
        xA = pt.prepare_batch_dev(dA, ref, "", w.ctx, cache)
        xG = pt.prepare_batch_dev(dG, ref, ".F", w.ctx, cache)
        torch.cuda.synchronize()
        t_prep.append(time.perf_counter() - t0)
        if os.environ.get("WCX_PROFILE_CONFIG5") and _ == runs - 1:      # dev aid: host profile -> stderr
            import cProfile
            import pstats
            pr = cProfile.Profile()
            w.ctx.lib.wcx_debug_flags(w.ctx.h, 8)          # CBS stage laps -> stderr
            rows = pr.runcall(wd.predict_batch_dev, w.backend, A, G, xA, xG, w.rem, pt)
            w.ctx.lib.wcx_debug_flags(w.ctx.h, 0)
            pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(8)
        else:
            rows = wd.predict_batch_dev(w.backend, A, G, xA, xG, w.rem, pt)
        torch.cuda.synchronize()
        t_all.append(time.perf_counter() - t0)
    ms = {k_: w.ctx.kernel_ms(k_) for k_ in ("aut:normalize", "normalize", "cbs", "segment_z")}
    checked = None
    if verify:
        # one sample of the batch against the pinned NumPy oracle (predict_tools.py:94-142: three dependent
        # passes over every autosomal bin, ~15 s on one core): n exact, z / r / medians to 1e-9
        from oracle import wcx_oracle as O
        i_chk = 41
        bufs = w.backend._predict_full_bufs
        gz, gr, gn = (bufs["a"][j][i_chk].cpu().numpy() for j in range(3))
        gmed = bufs["med"][:2, i_chk].cpu().numpy()
        idx_h, dist_h = A["idx"].cpu().numpy(), A["dist"].cpu().numpy()
        mb = [int(v) for v in w.p["masked_bins_per_chr"]]
        cum = [int(v) for v in w.cum]
        t0 = time.perf_counter()
        cutoff = float(O.get_optimal_cutoff(dist_h, int(w.pargs.maskrepeats)))
        oz, orr, on, omlr, omz = O.normalize_repeat(xA[i_chk].cpu().numpy(), mb, cum, idx_h, dist_h, cutoff, 0, 0)
        with np.errstate(all="ignore"):
            ok_n = bool(np.array_equal(gn, on))
            ok_r = bool(np.allclose(gr, orr, rtol=1e-9, atol=0.0, equal_nan=True))
            ok_z = bool(np.allclose(gz, oz, rtol=1e-9, atol=1e-9, equal_nan=True))
            ok_m = bool(np.allclose(gmed, [omlr, omz], rtol=1e-9, atol=1e-12))
        checked = {"sample": i_chk, "bins": int(len(on)), "n_exact": ok_n, "r_1e-9": ok_r, "z_1e-9": ok_z,
                   "medians_1e-9": ok_m, "mismatches": int(not (ok_n and ok_r and ok_z and ok_m)),
                   "oracle_seconds": time.perf_counter() - t0,
                   "what": "autosomal normalize_repeat outputs of one sample of the batch against "
                           "oracle.wcx_oracle.normalize_repeat (all bins, three passes)"}
    B, BG, k = int(w.B), int(w.P["F"]["B"] - int(w.P["F"]["cum"][21])), int(w.k)
    nbytes = 3 * (B + BG) * (k * 4 + k // 8) + 3 * batch * (B + BG) * 32
    norm_ms = max(ms["aut:normalize"], 0.0) + max(ms["normalize"], 0.0)
    return {"workload": "BASELINE configs[4]: predict {} samples at {} kb (autosomes + gonosomes, merge, "
                        "CBS, segment z), one device".format(batch, w.args.binsize // 1000),
            "batch_s": min(t_all), "runs_s": t_all, "prep_s": min(t_prep), "samples_per_s": batch / min(t_all),
            "kernel_ms": ms, "segments": int(sum(len(r_) for r_ in rows)), "verified": checked,
            "normalize_roofline": {"ms": norm_ms, "algorithmic_bytes": int(nbytes),
                                   "achieved_GBs": nbytes / (norm_ms * 1e-3) / 1e9 if norm_ms > 0 else None,
                                   "frac_of_hbm_peak": nbytes / (norm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if norm_ms > 0 else None,
                                   "bytes": "3 passes x (A + gonosomal rows) x (k indexes + selection bits), read once "
                                            "per batch + 32 B per (sample, bin, pass)"}}


def e2e_cli_block(w, workdir=None):
    """BASELINE.json's literal metric: wall-clock of the CLI -- `newref` on the cohort's sample files
    (load, gender model, masks, PCA, A / F / M searches + null ratios, reference .npz written, QC) and
    `predict --bed` of one sample.  Sample files are written before the clock starts."""
    import random
    import shutil
    from concurrent.futures import ThreadPoolExecutor
    import tempfile
    from wisecondorx_amd import main as cli, npz_io
    workdir = workdir or tempfile.mkdtemp(prefix="wcx_bench_e2e_")
    samples, genders = w.co.cohort_corrected
    def raw(s_, g_):                     # undo gender_correct (males' gonosomal counts were doubled)
        return s_ if g_ != "M" else dict(s_, **{"23": s_["23"] // 2, "24": s_["24"] // 2})
    files = [os.path.join(workdir, "s{:03d}.npz".format(i)) for i in range(len(samples))]
    with ThreadPoolExecutor(16) as ex:
        list(ex.map(lambda a_: npz_io.save_sample(a_[0], raw(a_[1], a_[2]), w.args.binsize),
                    zip(files, samples, genders)))
    test_file = os.path.join(workdir, "test.npz")
    npz_io.save_sample(test_file, w.co.sample(9001, "F", cnv=[(3, 500, 500 + int(3e7 / w.args.binsize), 1.5)]),
                       w.args.binsize)
    ref_file = os.path.join(workdir, "ref.npz")
    random.seed(1)
    out = {"workload": "CLI newref ({} sample files, {} bp bins, refsize {}, --aligned-masks --yfrac 0.004: the "
                       "synthetic cohort's gonosomal passes drop an autosomal bin, which the default refuses to "
                       "write, and its Y fractions have no mixture minimum) + predict --bed of 1 sample, one "
                       "device".format(len(files), w.args.binsize, w.k)}
    try:
        t0 = time.perf_counter()
        cli.main(["--loglevel", "error", "newref"] + files + [
            ref_file, "--binsize", str(w.args.binsize), "--refsize", str(w.k), "--yfrac", "0.004",
            "--aligned-masks"])
        out["newref_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        cli.main(["--loglevel", "error", "predict", test_file, ref_file, os.path.join(workdir, "out"),
                  "--bed", "--seed", "1"])
        out["predict_s"] = time.perf_counter() - t0
        out["newref_plus_predict_s"] = out["newref_s"] + out["predict_s"]
        out["reference_npz_bytes"] = os.path.getsize(ref_file)
        out["segments"] = sum(1 for _ in open(os.path.join(workdir, "out_segments.bed"))) - 1
    except SystemExit as e:
        out["error"] = "CLI exited ({})".format(e.code)
    finally:
        shutil.rmtree(workdir, ignore_errors=True)
    return out


def prep_pipeline_block(w):
    """SURVEY 8 rows a1-a3 + f2, outside the timed step: the A pass's preparation from the cohort's integer
    counts resident in HBM (what the CLI runs: prep.DeviceCounts -> get_mask -> prepare_dev = depth
    normalisation + mask, fp64 Gram, the five eigen-pairs on the host, PCA correction, distance filter).
    Wall-clock with the device drained around each stage + the device timers of the Gram and the
    correction kernels.  (newref_tools.py:77-147, newref_control.py:38-58.)"""
    from wisecondorx_amd import prep
    samples, _ = w.co.cohort_corrected
    ctx = w.ctx
    out = {}
    best = None
    for _ in range(2):                       # (the second run: allocations and the eigen-solver warm)
        t = {}
        ctx.sync()
        t0 = time.perf_counter()
        dc = prep.DeviceCounts(ctx, samples)
        ctx.sync()
        t["counts_to_device_ms"] = 1e3 * (time.perf_counter() - t0)
        t0 = time.perf_counter()
        mask, bpc = dc.get_mask()
        ctx.sync()
        t["get_mask_ms"] = 1e3 * (time.perf_counter() - t0)
        t0 = time.perf_counter()
        p = prep.prepare_dev(dc, np.arange(len(samples)), "A", mask, bpc)
        ctx.sync()
        t["prepare_ms"] = 1e3 * (time.perf_counter() - t0)
        t["pca_gram_kernel_ms"] = ctx.kernel_ms("pca_gram")
        t["pca_apply_kernel_ms"] = ctx.kernel_ms("pca_apply")
        dc.close()
        ctx.lib.wcx_pca_end(ctx.h)
        t["prep_pipeline_ms"] = t["get_mask_ms"] + t["prepare_ms"]
        if best is None or t["prep_pipeline_ms"] < best["prep_pipeline_ms"]:
            best = t
    out.update(best)
    out["bins_kept"] = int(p["masked_bins_per_chr_cum"][-1])
    out["what"] = "A pass of the bench cohort from int32 counts in HBM: mask (a1), depth normalisation + Gram + " \
                  "host eigen-solve + correction (a1, a2 / f2), PCA-distance filter (a3; a second Gram + " \
                  "correction when it fires); wall-clock, device drained; counts_to_device_ms = the host " \
                  "layout + upload of the cohort's counts, not part of prep_pipeline_ms"
    return out


def cpu_baseline_predict(w, budget_bins=2048):
    """CPU leg for the predict path: the pinned NumPy oracle's normalize_once (predict_tools.py:111-142)
    for a block of target bins (each against its full reference row), the three passes of
    normalize_repeat, 1 core; extrapolated linearly in bins (the cost per bin is constant)."""
    from oracle import wcx_oracle as O
    A = w.last_ref["A"]
    idx = A["idx"].cpu().numpy()
    dist = A["dist"].cpu().numpy()
    x = w.d_xA.cpu().numpy()
    mb = [int(v) for v in w.p["masked_bins_per_chr"]]
    cum = [int(v) for v in w.cum]
    cutoff = float(O.get_optimal_cutoff(dist[::16], 5))
    lo = cum[2]                                     # a block inside chromosome 4
    t0 = time.perf_counter()
    copy = np.copy(x)
    for _ in range(3):
        z, r, n = O.normalize_once(x, copy, mb, cum, idx, dist, cutoff, 0, 0, row_range=(lo, lo + budget_bins))
        with np.errstate(all="ignore"):
            copy[np.abs(z) >= 3] = -1
    dt = time.perf_counter() - t0
    per_bin = dt / budget_bins
    return {"value": 1.0 / per_bin, "unit": "bins/s (3 normalisation passes, k refs each)", "cores": 1,
            "kind": "port", "sample": "{} of {} autosomal bins, one sample".format(budget_bins, len(x)),
            "seconds": dt, "extrapolated_s_per_sample": per_bin * len(x),
            "gpu_ms_per_sample_A_plus_gonosomes": w.mean_ms("normalize")}


def summarize_collectives(log):
    """Per collective op of dist.collective_report(): calls, payload bytes, device ms (events on the
    current stream around the blocking collective) and host ms."""
    out = {}
    for e in log:
        d = out.setdefault(e["op"], {"calls": 0, "bytes": 0, "ms": 0.0, "host_ms": 0.0, "backend": e["backend"]})
        d["calls"] += 1
        d["bytes"] += e["bytes"]
        d["ms"] += e.get("ms", 0.0)
        d["host_ms"] += e["host_ms"]
    return out


def rccl_world1_block(w, torch, dev):
    """The multi-GPU code path on the ONE GPU this run has: a process group of one rank on backend
    "nccl" (= RCCL), the world == 1 short-circuits of dist.py off (WCX_FORCE_COLLECTIVES=1), one extra
    untimed step -- A pass through dist.newref_sym_sharded (all-gather of X, all-to-all of the hit
    records), F / M passes through newref_gonosomal_sharded, gather_reference3 (padded all-gather +
    compaction) -- and its tables compared bit for bit with those of the last timed step.  Says that the
    collectives EXECUTE and order correctly against the library's kernels; no scaling is measured."""
    import socket
    import torch.distributed as dist
    wd = w.wd
    out = {"ok": False, "what": "one untimed step with every collective of dist.py forced through a 1-rank "
                                "process group on backend nccl (RCCL); tables == the last timed step's, bitwise"}
    keep = {tag: {k_: v.clone() for k_, v in w.last_ref[tag].items() if k_ != "cum"} for tag in w.last_ref}
    env_old = {k_: os.environ.get(k_) for k_ in ("WCX_FORCE_COLLECTIVES", "WCX_SYM_SHARD_MIN", "MASTER_ADDR",
                                                 "MASTER_PORT")}
    side_, w.side = w.side, {}
    try:
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        os.environ.update({"WCX_FORCE_COLLECTIVES": "1", "WCX_SYM_SHARD_MIN": "1", "MASTER_ADDR": "127.0.0.1",
                           "MASTER_PORT": str(port)})
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        try:
            wd.COLLECTIVE_LOG = []
            wd.newref_sym_sharded.last_records = None
            w.step(False)
            torch.cuda.synchronize()
            log = wd.collective_report()
            same, diffs = True, {}
            for tag in keep:
                for k_, v in keep[tag].items():
                    g = w.last_ref[tag][k_]
                    eq = g.shape == v.shape and bool(
                        torch.equal(g.contiguous().view(torch.uint8), v.contiguous().view(torch.uint8)))
                    if not eq:
                        if g.shape == v.shape:       # rows that differ (NaN == NaN bitwise)
                            gb = g.contiguous().view(torch.uint8).reshape(g.shape[0], -1)
                            vb = v.contiguous().view(torch.uint8).reshape(v.shape[0], -1)
                            rows_ = torch.nonzero((gb != vb).any(dim=1)).flatten()
                            diffs["{}:{}".format(tag, k_)] = {"rows": int(rows_.numel()),
                                                              "first": [int(x) for x in rows_[:4].tolist()]}
                        else:
                            diffs["{}:{}".format(tag, k_)] = {"shape": [list(g.shape), list(v.shape)]}
                    same = same and eq
            if diffs:
                out["differences"] = diffs
            out.update({"ok": bool(same and len(log) > 0), "tables_equal": bool(same),
                        "records_sent_received": wd.newref_sym_sharded.last_records,
                        "collectives": summarize_collectives(log)})
        finally:
            wd.COLLECTIVE_LOG = None
            dist.destroy_process_group()
    except Exception as e:          # (reported, never fatal for the bench line)
        out["error"] = "{}: {}".format(type(e).__name__, e)
    finally:
        w.side = side_
        for k_, v in env_old.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v
    return out


def run_steps(w, steps, warmup, spinup, barrier):
    # Untimed spin-up (setup, not one of the contract's warm-up steps; reported in config): the
    # first process on an idle box otherwise measures the clock ramp.  A FIXED number of steps:
    # every rank must issue the same collectives.
    for _ in range(spinup):
        w.step(False)
    for _ in range(warmup):
        w.step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        w.step(True)
    barrier()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--binsize", type=int, default=15000)
    ap.add_argument("--samples", type=int, default=500)
    ap.add_argument("--refsize", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the S=100 block")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-run oracle check")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip config 5, the 100 kb config and the CLI end-to-end block")
    ap.add_argument("--replicas", action="store_true",
                    help="throughput mode: every rank builds its OWN whole reference + predict (N cohorts "
                         "side by side, no collective on the data path; 'scaling': 'weak') instead of "
                         "row-sharding ONE reference over the ranks (the default, 'strong')")
    ap.add_argument("--debug-flags", type=int, default=0, help="profiling ablations (invalid results)")
    ap.add_argument("--concurrent-passes", type=int, default=1,
                    help="1 (default, one GPU) = the F / M passes run on their own contexts and streams and "
                         "start when the A pass's sweep is done (wcx_sweep_event): their MFMA sweeps beside "
                         "A's L2-bound refine (15 kb x 500: step 72.4 -> 71.3 ms, x 100: 39.8 -> 37.6 ms); "
                         "0 = one pass after the other, clean per-pass kernel times")
    args = ap.parse_args()

    # The contract is ONE line on stdout.  Libraries print there too (RCCL's version banner at the first
    # process group, from C stdio): the real stdout is put aside and file descriptor 1 points at stderr
    # for the whole run; the JSON line goes to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("warning: --gpus {} but WORLD_SIZE {}".format(args.gpus, world), file=sys.stderr)
    # WCX_DIST_BACKEND=gloo is for tests only (several ranks sharing one device; RCCL needs one
    # device per rank)
    backend_name = os.environ.get("WCX_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend_name != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend_name == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend_name)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    part_rank, part_world = (0, 1) if args.replicas else (rank, world)
    w = Workload(args, args.samples, torch, dev, dev_index, part_rank, part_world)
    dt = run_steps(w, args.steps, args.warmup, SPINUP_STEPS, barrier)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend_name == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3

    ranks_info = None
    if world > 1 and not args.replicas and not args.debug_flags:
        # one extra untimed step with the collectives logged (events around each blocking collective):
        # per rank its rows, the hit records it sent / received, and what each collective cost
        from wisecondorx_amd import dist as wd_
        from wisecondorx_amd.newref_tools import _get_part as gp_
        wd_.COLLECTIVE_LOG = []
        wd_.newref_sym_sharded.last_records = None
        w.step(False)
        torch.cuda.synchronize()
        log_ = wd_.collective_report()
        wd_.COLLECTIVE_LOG = None
        rb_, re_ = gp_(rank, world, int(w.B))
        st_ = w.ctx.topk_stats()
        info_ = {"rank": rank, "rows_A": int(re_ - rb_), "pairs_A": int(st_.get("pairs", 0)),
                 "records_sent_received": wd_.newref_sym_sharded.last_records,
                 "collectives": summarize_collectives(log_)}
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, info_)
        barrier()

    def sequential_kernel_times(w_):
        """Per-kernel times (roofline, hbm_kernels) come from three EXTRA, untimed steps with the passes
        one after another: in the timed steps the F / M passes overlap the A pass's refine."""
        if not w_.side:
            return False
        side_, w_.side = w_.side, {}
        for k_ in w_.ms:
            w_.ms[k_] = []
        for _ in range(3):
            w_.step(True)
        barrier()
        w_.side = side_
        return True
    seq = sequential_kernel_times(w)
    roofline, screen_ms = w.roofline()
    if seq:
        roofline["kernel_times_from"] = "three extra untimed steps with the A / F / M passes one after " \
                                        "another (the timed steps overlap the F / M passes with A's refine)"

    # HBM traffic of the dominant kernel comes from rocprofv3 PMC passes (bench.py cannot read
    # counters itself); only valid for the kernel sources it was recorded with
    def add_traffic(rf, S, B):
        # what `frac` is: ALGORITHMIC flop (2 S per ordered pair) over the sweep's time -- the symmetric
        # sweep executes about half of them, so frac is a speed-up-adjusted figure, not the matrix
        # pipe's utilisation; frac_executed and (with the PMC file) mfma_busy_frac are the hardware's
        rf["frac_note"] = "algorithmic flop (2 S P) / sweep time / dense f16 peak; the hardware-side figures " \
                          "are frac_executed (MFMA flop really issued) and mfma_busy_frac (PMC)"
        rf["frac_executed"] = rf["executed_tflops"] / F16_MFMA_PEAK_TFLOPS
        algo_bytes = B * S * 8 + B * args.refsize * 12       # X once + idx / dist out
        rf["algorithmic_bytes"] = int(algo_bytes)
        if world != 1:
            return
        for path in TRAFFIC_JSONS:
            if not os.path.exists(path):
                continue
            tj = json.load(open(path))
            key = "S{}".format(S)
            if tj.get("kernel_sha") == screen_source_sha() and key in tj.get("workloads", {}) \
                    and (args.binsize, args.refsize) == (15000, 300):
                e = tj["workloads"][key]
                rf["traffic"] = e["fetch_bytes_per_sweep_corrected_x2"] + e["write_bytes_per_sweep"]
                rf["traffic_over_algorithmic"] = rf["traffic"] / algo_bytes
                if e.get("mfma_busy_cycles_per_sweep"):
                    rf["mfma_busy_frac"] = e["mfma_busy_cycles_per_sweep"] / (
                        1024 * rf["kernel_ms"] * 1e-3 * SIMD_CLOCK_HZ)
                    rf["mfma_busy_note"] = "SQ_VALU_MFMA_BUSY_CYCLES per sweep / (1 024 SIMDs x kernel_ms x " \
                                           "2.4 GHz)"
                rf["traffic_source"] = "{} (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES; " \
                                       "per screen sweep; kernel_sha {})".format(
                                           os.path.relpath(path, ROOT), tj["kernel_sha"])
                return
    if screen_ms >= 0:
        add_traffic(roofline, w.S, w.B)

    out = {
        "metric": "newref+predict throughput @{}kb bins (bins x refs per second)".format(
            args.binsize // 1000),
        "value": (world if args.replicas else 1) * w.pairs_total / (ms_per_step * 1e-3),
        "unit": "bin-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak" if args.replicas else "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "newref {} kb bins, {} samples, refsize={}: A pass (B={} masked autosomal "
                               "bins x S={}) + F pass ({} chrX rows x S={}) + M pass ({} chrX/Y rows x "
                               "S={}), each search + null ratios + gather of the reference; + predict of "
                               "1 sample (cut-off, weights, 3 normalisation passes for autosomes and "
                               "gonosomes, merge, post-processing, CBS of 23 chromosomes, segment z).  The step "
                               "starts from the prepared matrices resident in HBM: masks, depth normalisation, "
                               "PCA correction + distance filter (newref_tools.py:77-147, newref_control.py:38-58) "
                               "and all file I/O are OUTSIDE it -- their device time is `prep_pipeline`, the "
                               "whole CLI `e2e_cli`".format(
                                   args.binsize // 1000, w.S, w.k,
                                   w.B, w.S, w.P["F"]["B"] - int(w.P["F"]["cum"][21]), w.P["F"]["S"],
                                   w.P["M"]["B"] - int(w.P["M"]["cum"][21]), w.P["M"]["S"]),
                   "bins": int(w.B), "samples": int(w.S), "refsize": int(w.k),
                   "pairs": w.pairs_total, "bin_samples_per_s": w.B * w.S / (ms_per_step * 1e-3),
                   "spinup_steps_untimed": SPINUP_STEPS, "predict_segments": w.n_segments,
                   "precision": "indices and distances bit-identical to the reference's fp64 path; the "
                                "fp16 MFMA product is only a rigorously bounded pre-filter, every "
                                "kept pair is re-evaluated in sequential fp64",
                   "partition": "target rows x{} (A: _get_part over all rows; F / M: over their gonosomal "
                                "rows): per pass one all-gather(X) for the search and one all-gather of "
                                "the finished row blocks; predict replicated".format(world)
                                if not args.replicas else
                                "{} independent replicas (one whole reference + predict per rank), no "
                                "collective on the data path".format(world)},
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_secondary and not args.debug_flags \
            and args.samples != 100:
        w2 = Workload(args, 100, torch, dev, dev_index, rank, world)
        dt2 = run_steps(w2, args.steps, args.warmup, SPINUP_STEPS, barrier)
        sequential_kernel_times(w2)
        r2, sm2 = w2.roofline()
        if sm2 >= 0:
            add_traffic(r2, w2.S, w2.B)
        out["secondary"] = {"workload": "BASELINE configs[2]: 15 kb x 100 samples, same step",
                            "ms_per_step": dt2 / args.steps * 1e3,
                            "value": w2.pairs_total / (dt2 / args.steps),
                            "roofline": {k_: r2[k_] for k_ in r2
                                         if k_ not in ("kernel", "attainable_note",
                                                       "fallback_rows_per_step")}}
        if not args.no_verify:
            out["secondary"]["verified"] = verify_rows(w2, n_blocks=40, rows_per_block=16, gon_blocks=8)
    if ranks_info is not None:
        out["ranks"] = ranks_info
    if rank == 0 and world == 1 and not args.debug_flags:
        out["hbm_kernels"] = hbm_kernels(w, roofline)
    if rank == 0 and not args.debug_flags and not args.no_verify:
        out["verified"] = verify_rows(w)
    if rank == 0 and world == 1 and not args.debug_flags and not args.no_extras and backend_name == "nccl":
        out["rccl_world1"] = rccl_world1_block(w, torch, dev)
        out["extra"] = {"rccl_world1_ok": out["rccl_world1"]["ok"]}
    if rank == 0 and world == 1 and not args.debug_flags and not args.no_extras:
        # the other BASELINE configs and the literal wall-clock metric, outside the timed region
        out["config5"] = config5_block(w, verify=not args.no_verify)
        try:
            out["prep_pipeline"] = prep_pipeline_block(w)
        except Exception as e:             # (reported, never fatal for the bench line)
            out["prep_pipeline"] = {"error": "{}: {}".format(type(e).__name__, e)}
        a2 = argparse.Namespace(**dict(vars(args), binsize=100000))
        w3 = Workload(a2, 100, torch, dev, dev_index, rank, world)
        dt3 = run_steps(w3, args.steps, args.warmup, SPINUP_STEPS, barrier)
        sequential_kernel_times(w3)
        r3, _ = w3.roofline()
        out["config2_100kb"] = {"workload": "BASELINE configs[1]: 100 kb x 100 samples, same step",
                                "ms_per_step": dt3 / args.steps * 1e3, "value": w3.pairs_total / (dt3 / args.steps),
                                "bins": int(w3.B), "screen_ms": r3.get("kernel_ms"), "refine_ms": r3.get("refine_ms"),
                                "roofline_frac": r3.get("frac"), "null_ratios_ms": r3.get("null_ratios_ms"),
                                "pre_ms": r3.get("pre_ms"), "appends": r3.get("appends"),
                                "fallback_rows": r3.get("fallback_rows"),
                                "gonosomal_passes": r3.get("gonosomal_passes")}
        if not args.no_verify:
            out["config2_100kb"]["verified"] = verify_rows(w3, n_blocks=40, rows_per_block=16, gon_blocks=8)
        # the CLI's wall-clock at the reference's DEFAULT bin size (main.py:377-380) -- the common
        # production shape -- beside the headline 15 kb one
        out["e2e_cli_100kb"] = e2e_cli_block(w3)
        del w3
        out["e2e_cli"] = e2e_cli_block(w)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(w.Xs_host, w.cum, w.k)
        out["cpu_baseline"]["predict"] = cpu_baseline_predict(w)
    if world > 1:
        barrier()                   # (rank 0 verifies after the timed region: nobody tears the group down under it)
        dist.destroy_process_group()
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)           # C stdio buffers (RCCL's banner) -> stderr, not after the JSON
    except Exception:
        pass
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
