#!/usr/bin/env python3
"""Per-rank compute of an N-GPU newref A pass with the ROW-SHARDED SYMMETRIC sweep
(dist.newref_sym_sharded), measured on ONE device: for N = 1, 2, 4, 8 every rank's phase 1 (prep +
thresholds of all rows + its share of the tile pairs + record bucketing) runs in turn -- on real
hardware they run side by side --, the records for one median rank are collected like the all-to-all
would deliver them, and that rank's phase 2 (lists, final cut, refine, null ratios) is timed.
Projected wall of the rank = its phase 1 + the exchange (bytes at 300 GB/s, stated) + its phase 2.
Beside it: the same rank's shard with the one-directional sweep (be.search on the row range), the
round-1..4 form.  usage: bench_shard_sym.py [S]"""
import json
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
    from wisecondorx_amd import dist as wd
    from wisecondorx_amd.newref_tools import _get_part
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    k = 300
    co, p, _ = bench.make_workload(15000, S)
    X = p["X"]
    cum = np.asarray(p["masked_bins_per_chr_cum"], dtype=np.int64)
    B = int(cum[-1])
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    be = wd.GpuBackend(ctx)
    d_Xs = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)
    ids = np.arange(min(S, 100), dtype=np.int32)
    out = {"S": S, "B": B, "k": k}
    for n in (1, 2, 4, 8):
        me = {1: 0, 2: 1, 4: 1, 8: 3}[n]
        bounds = [_get_part(r, n, B)[0] for r in range(n)] + [B]
        r0, r1 = bounds[me], bounds[me + 1]
        rows = r1 - r0
        d_idx = torch.empty((rows, k), dtype=torch.int32, device=dev)
        d_dist = torch.empty((rows, k), dtype=torch.float64, device=dev)
        d_nr = torch.empty((rows, len(ids)), dtype=torch.float64, device=dev)
        entry = {"rows": rows}
        # one-directional shard (round 1..4)
        for _ in range(3):
            ctx.sync()
            t0 = time.perf_counter()
            be.search(d_Xs, B, S, cum, r0, r1, k, ids, d_idx, d_dist, d_nr)
            ctx.sync()
            wall = 1e3 * (time.perf_counter() - t0)
        entry["one_directional"] = {"shard_wall_ms": round(wall, 3),
                                    "screen_ms": round(ctx.kernel_ms("topk_screen"), 3),
                                    "refine_ms": round(ctx.kernel_ms("topk_refine"), 3),
                                    "null_ratios_ms": round(ctx.kernel_ms("null_ratios"), 3)}
        ref_idx = d_idx.clone()
        if n == 1:
            out["N1"] = entry
            continue
        # symmetric shards: every rank's phase 1 in turn, the records for rank `me` collected
        best = None
        for rep in range(2):
            mine, p1 = [], {}
            for r in range(n):
                ctx.sync()
                t0 = time.perf_counter()
                counts = be.sym_sweep(d_Xs, B, S, cum, k, r, n, bounds, ids)
                if counts is None:
                    raise SystemExit("the library has no symmetric sweep for this shape")
                send = torch.empty((int(sum(counts)), 4), dtype=torch.int32, device=dev)
                be.sym_records(send)
                ctx.sync()
                p1[r] = (1e3 * (time.perf_counter() - t0), int(sum(counts)),
                         ctx.kernel_ms("topk_prep"), ctx.kernel_ms("topk_pre"), ctx.kernel_ms("topk_screen"))
                lo = int(sum(counts[:me]))
                mine.append(send[lo:lo + counts[me]].clone())
                if r != me:
                    # (phase 2 of the other ranks is not run: close their state by a finish on no records
                    #  would flag rows; instead the next sweep simply overwrites the state)
                    pass
            # rank me's phase 1 state is gone (later sweeps overwrote it): run it again, last
            counts = be.sym_sweep(d_Xs, B, S, cum, k, me, n, bounds, ids)
            send = torch.empty((int(sum(counts)), 4), dtype=torch.int32, device=dev)
            be.sym_records(send)
            recv = torch.cat(mine)
            ctx.sync()
            t0 = time.perf_counter()
            be.sym_finish(recv, d_Xs, B, S, cum, r0, r1, k, ids, d_idx, d_dist, d_nr)
            ctx.sync()
            p2 = 1e3 * (time.perf_counter() - t0)
            ok = bool(torch.equal(d_idx, ref_idx))
            sent_bytes = 16 * (p1[me][1] - counts[me])
            a2a_ms = 1e3 * sent_bytes / 300e9
            cur = {"phase1_ms": round(p1[me][0], 3), "phase1_all_ranks_ms": [round(p1[r][0], 3) for r in range(n)],
                   "prep_ms": round(p1[me][2], 3), "thresholds_ms": round(p1[me][3], 3),
                   "screen_ms": round(p1[me][4], 3), "records_sent": p1[me][1], "records_received": int(recv.shape[0]),
                   "all_to_all_ms_at_300GBs": round(a2a_ms, 3), "phase2_ms": round(p2, 3),
                   "cut_ms": round(ctx.kernel_ms("topk_cut"), 3), "refine_ms": round(ctx.kernel_ms("topk_refine"), 3),
                   "null_ratios_ms": round(ctx.kernel_ms("null_ratios"), 3),
                   "fallback_rows": ctx.topk_stats()["fallback_rows"],
                   "identical_to_one_directional_shard": ok,
                   "projected_wall_ms": round(p1[me][0] + a2a_ms + p2, 3)}
            if best is None or cur["projected_wall_ms"] < best["projected_wall_ms"]:
                best = cur
        entry["symmetric"] = best
        out["N{}".format(n)] = entry
    print(json.dumps(out))


if __name__ == "__main__":
    main()
