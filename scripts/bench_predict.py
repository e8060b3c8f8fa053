#!/usr/bin/env python3
"""Config 5 evidence: predict a batch of samples at 15 kb on ONE MI355X (samples would be striped
over the 8 GPUs of a node; there is no collective on this path).  Times: reference upload,
cut-off, weights, batched 3-pass normalisation, host post-processing, GPU CBS, segment z."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--binsize", type=int, default=15000)
    ap.add_argument("--samples", type=int, default=100)
    ap.add_argument("--batch", type=int, default=96)
    ap.add_argument("--cbs-samples", type=int, default=3)
    ap.add_argument("--streams", type=int, default=4, help="host threads / HIP streams for the CBS stage")
    a = ap.parse_args()
    import bench
    from wisecondorx_amd import _lib, newref_tools, predict_tools as pt
    co, p, _ = bench.make_workload(a.binsize, a.samples)
    X = p["X"]
    cum = [int(v) for v in p["masked_bins_per_chr_cum"]]
    ctx = _lib.default_context(0)
    t = time.perf_counter()
    idx, dist = newref_tools.get_ref_for_rows(X, cum, 300, 0, cum[-1], ctx)
    nr = newref_tools.get_null_ratios(X, idx, 0, cum[-1], list(range(min(a.samples, 100))), ctx)
    t_newref = time.perf_counter() - t
    ref = dict(p)
    ref.update({"indexes": idx, "distances": dist, "null_ratios": nr})
    tests = [co.sample(5000 + i, "F", cnv=[(1 + i % 22, 200, 200 + 2000, 1.5)]) for i in range(a.batch)]
    t = time.perf_counter()
    xs = np.stack([pt.project_pc(pt.coverage_normalize_and_mask(s, ref, ""), ref, "") for s in tests])
    t_host_prep = time.perf_counter() - t
    cache = {}
    t = time.perf_counter(); pt._dev(ref, "", cache); ctx.sync(); t_upload = time.perf_counter() - t
    t = time.perf_counter(); cutoff = pt.get_optimal_cutoff(ref, 5, cache); t_cut = time.perf_counter() - t
    t = time.perf_counter(); w = pt.get_weights(ref, "", cache); t_w = time.perf_counter() - t
    t = time.perf_counter()
    z, r, n, mlr, mz = pt.normalize_repeat_batch(xs, ref, cutoff, 0, 0, "", cache)
    t_norm = time.perf_counter() - t
    norm_kernel_ms = ctx.kernel_ms("normalize")
    args = argparse.Namespace(minrefbins=150, alpha=1e-4, seed=1)
    rem = {"args": args, "mask": ref["mask"], "bins_per_chr": ref["bins_per_chr"],
           "binsize": a.binsize, "ref_gender": "F"}
    t_post = t_cbs = t_segz = 0.0
    n_seg = []
    t = time.perf_counter()
    off = np.concatenate(([0], np.cumsum(ref["bins_per_chr"]))).astype(int)
    nr_full = pt.inflate_results(nr, rem)
    pt.attach_null_matrix([nr_full[off[c]:off[c + 1]] for c in range(len(off) - 1)], ctx)
    t_attach = time.perf_counter() - t
    for i in range(min(a.cbs_samples, a.batch)):
        t = time.perf_counter()
        res = {"results_r": r[i], "results_z": z[i] - mz[i], "results_w": w / np.nanmean(w)}
        for k in res:
            res[k] = pt.get_post_processed_result(args, res[k], n[i], rem)
        res["results_nr"] = pt.ATTACHED
        pt.log_trans(res, mlr[i])
        t_post += time.perf_counter() - t
        t = time.perf_counter()
        segs = pt.run_cbs(res, "A", args.alpha, a.binsize, args.seed, ctx)
        t_cbs += time.perf_counter() - t
        t = time.perf_counter()
        pt.get_z_score(segs, res, ctx)
        t_segz += time.perf_counter() - t
        n_seg.append(len(segs))
    m = max(1, min(a.cbs_samples, a.batch))
    # the same stage for the WHOLE batch, samples striped over --streams contexts (own streams)
    ctxs = [ctx] + [_lib.Context(0) for _ in range(a.streams - 1)]
    for c in ctxs[1:]:
        pt.attach_null_matrix([nr_full[off[c2]:off[c2 + 1]] for c2 in range(len(off) - 1)], c)

    w_scaled = w / np.nanmean(w)

    def post(i):       # fused minrefbins / inflate / log2 pass (== the step-by-step sequence above)
        with np.errstate(all="ignore"):
            res = pt.post_process_fused(args, r[i], z[i] - mz[i], w_scaled, n[i], mlr[i], rem)
        res["results_nr"] = pt.ATTACHED
        return res
    t = time.perf_counter()
    all_segs = pt.segment_batch(list(range(a.batch)), rem, ctxs, post=post)
    t_batch = time.perf_counter() - t
    # host prep (coverage vector + PCA projection) of the batch on host threads
    from concurrent.futures import ThreadPoolExecutor
    t = time.perf_counter()
    with ThreadPoolExecutor(max_workers=a.streams) as ex:
        xs_t = list(ex.map(lambda smp: pt.project_pc(pt.coverage_normalize_and_mask(smp, ref, ""), ref, ""),
                           tests))
    t_host_prep_threaded = time.perf_counter() - t
    assert np.array_equal(np.stack(xs_t), xs)
    # ---- the same batch DEVICE-RESIDENT end to end (dist.predict_batch_dev): normalisation outputs
    # stay in HBM through the merge / post-processing kernel, the batched CBS and the batched segment z
    import torch
    from wisecondorx_amd import dist as wd
    dev = torch.device("cuda", 0)
    ctx_t = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    be = wd.GpuBackend(ctx_t)
    tt = lambda arr: torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
    A = {"idx": tt(idx), "dist": tt(dist), "nr": tt(nr), "cum": cum}
    rem_dev = dict(rem, args=argparse.Namespace(minrefbins=150, alpha=1e-4, seed=1, maskrepeats=5))
    t_dev, t_prep_dev = [], []
    prep_cache = {}
    for _ in range(3):
        t = time.perf_counter()
        # sample preparation on the device too: counts laid out over the reference's bins (host
        # memcpys), one H2D, coverage normalisation + mask + PCA projection in three kernels
        d_counts = torch.from_numpy(pt.sample_counts_matrix(tests, ref, "")).to(dev)
        d_x = pt.prepare_batch_dev(d_counts, ref, "", ctx_t, prep_cache)
        ctx_t.sync()
        t_prep_dev.append(time.perf_counter() - t)
        rows_dev = wd.predict_batch_dev(be, A, None, d_x, None, rem_dev, pt)
        t_dev.append(time.perf_counter() - t)
    x_err = float(np.max(np.abs(d_x.cpu().numpy() / xs - 1.0)))
    same = [[r_[:3] for r_ in rows_dev[i]] == [r_[:3] for r_ in all_segs[i]] for i in range(a.batch)]
    dev_ms = {k_: ctx_t.kernel_ms(k_) for k_ in ("aut:normalize", "cbs", "segment_z")}
    print(json.dumps({
        "workload": "predict batch: {} samples, {} kb bins, B={}, k=300".format(a.batch, a.binsize // 1000, cum[-1]),
        "newref_host_api_s": t_newref, "host_prep_per_sample_ms": 1e3 * t_host_prep / a.batch,
        "ref_upload_s": t_upload, "null_matrix_attach_s": t_attach, "cutoff_ms": 1e3 * t_cut, "weights_ms": 1e3 * t_w,
        "normalize_batch_s": t_norm, "normalize_kernels_ms": norm_kernel_ms,
        "normalize_per_sample_ms": 1e3 * t_norm / a.batch,
        "postprocess_per_sample_ms": 1e3 * t_post / m, "cbs_per_sample_s": t_cbs / m,
        "segment_z_per_sample_ms": 1e3 * t_segz / m, "segments": n_seg,
        "streams": a.streams, "post_cbs_segz_whole_batch_s": t_batch,
        "post_cbs_segz_per_sample_ms_threaded": 1e3 * t_batch / a.batch,
        "host_prep_threaded_per_sample_ms": 1e3 * t_host_prep_threaded / a.batch,
        "whole_batch_s": t_host_prep_threaded + t_norm + t_batch,
        "device_resident_batch_s": min(t_dev), "device_resident_runs_s": t_dev,
        "device_resident_kernel_ms": dev_ms, "device_resident_same_segments": int(sum(same)),
        "device_prep_s": min(t_prep_dev), "device_prep_max_rel_err_vs_host": x_err,
        "whole_batch_device_resident_s": min(t_dev),
        "batch_segments": int(sum(len(x) for x in all_segs))}))


if __name__ == "__main__":
    main()
