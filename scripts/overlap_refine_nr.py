#!/usr/bin/env python3
"""Would null ratios BESIDE the refine pay?  The A pass's search on the main context and the null ratios of
(the previous step's) index table on a second context / stream that waits for the search's sweep event:
wall time of both against one after the other.  An experiment, not a product path.
usage: overlap_refine_nr.py [samples]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from wisecondorx_amd import _lib
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    k = 300
    co, p, _ = bench.make_workload(15000, S)
    X = p["X"]
    cum = np.asarray(p["masked_bins_per_chr_cum"], dtype=np.int64)
    B = int(cum[-1])
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    st2 = torch.cuda.Stream(device=dev)
    ctx2 = _lib.Context(0, st2.cuda_stream)
    ev = _lib.vp()
    _lib.check(ctx.lib.wcx_sweep_event(ctx.h, _lib.C.byref(ev)))
    d_Xs = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)
    ids = np.arange(min(S, 100), dtype=np.int32)
    _, ids_p = _lib.i32_array(ids)
    _, cum_p = _lib.i64_array(cum)
    d_idx = torch.empty((B, k), dtype=torch.int32, device=dev)
    d_idx2 = torch.empty((B, k), dtype=torch.int32, device=dev)
    d_dist = torch.empty((B, k), dtype=torch.float64, device=dev)
    d_nr = torch.empty((B, len(ids)), dtype=torch.float64, device=dev)
    d_nr2 = torch.empty((B, len(ids)), dtype=torch.float64, device=dev)
    lib = ctx.lib

    def topk(c, out_idx):
        _lib.check(lib.wcx_newref_topk_dev(c.h, d_Xs.data_ptr(), B, S, cum_p, len(cum), 0, B, k, 0,
                                           out_idx.data_ptr(), d_dist.data_ptr()))

    def nr(c, in_idx, out):
        _lib.check(lib.wcx_null_ratios_dev(c.h, d_Xs.data_ptr(), B, S, in_idx.data_ptr(), 0, B, k, ids_p,
                                           len(ids), out.data_ptr()))
    topk(ctx, d_idx)
    nr(ctx, d_idx, d_nr)
    torch.cuda.synchronize()
    d_idx2.copy_(d_idx)
    res = {}
    for mode in ("sequential", "overlapped", "sequential", "overlapped"):
        ts = []
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            topk(ctx, d_idx)
            if mode == "sequential":
                nr(ctx, d_idx2, d_nr2)
            else:
                with torch.cuda.stream(st2):
                    _lib.check(lib.wcx_wait_event(ctx2.h, ev))
                    nr(ctx2, d_idx2, d_nr2)
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        res.setdefault(mode, []).append(float(np.median(ts)))
        print(mode, "%.2f ms" % np.median(ts), "refine %.2f" % ctx.kernel_ms("topk_refine"),
              "screen %.2f" % ctx.kernel_ms("topk_screen"))
    assert torch.equal(d_nr, d_nr2)


if __name__ == "__main__":
    main()
