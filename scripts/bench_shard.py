#!/usr/bin/env python3
"""Per-rank compute of an N-GPU newref, measured on ONE device: rank r = 3 (N = 8), 1 (N = 2, 4), 0
(N = 1) builds its row shard of the 15 kb problem (search + null ratios) for N = 1, 2, 4, 8 and several
candidate-segment counts (WCX_SCREEN_SEGMENTS); wall time of the whole shard build incl. the ranking
on the auxiliary stream.  This is what the 1 -> 8 GPU curve can at best look like (the all-gathers come
on top): DESIGN.md section 5."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from wisecondorx_amd import _lib
    from wisecondorx_amd import dist as wd
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    co, p, _ = bench.make_workload(15000, S)
    X = p["X"]
    cum = np.asarray(p["masked_bins_per_chr_cum"], dtype=np.int64)
    B = int(cum[-1])
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    be = wd.GpuBackend(ctx)
    d_Xs = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)
    ids = np.arange(min(S, 100), dtype=np.int32)
    out = {}
    import time
    from wisecondorx_amd.newref_tools import _get_part
    for n in (1, 2, 4, 8):
        r0, r1 = _get_part(min(n - 1, {1: 0, 2: 1, 4: 1, 8: 3}[n]), n, B)
        rows = r1 - r0
        d_idx = torch.empty((rows, 300), dtype=torch.int32, device=dev)
        d_dist = torch.empty((rows, 300), dtype=torch.float64, device=dev)
        d_nr = torch.empty((rows, len(ids)), dtype=torch.float64, device=dev)
        for seg in ("auto", "1", "4", "8"):
            if seg == "auto":
                os.environ.pop("WCX_SCREEN_SEGMENTS", None)
            else:
                os.environ["WCX_SCREEN_SEGMENTS"] = seg
            for _ in range(3):
                ctx.sync()
                t0 = time.perf_counter()
                be.search(d_Xs, B, S, cum, r0, r1, 300, ids, d_idx, d_dist, d_nr)
                ctx.sync()
                wall = 1e3 * (time.perf_counter() - t0)
            out["N{}_seg{}".format(n, seg)] = {
                "rows": rows, "shard_wall_ms": round(wall, 3),
                "fallback_rows": ctx.topk_stats()["fallback_rows"],
                "screen_ms": round(ctx.kernel_ms("topk_screen"), 3),
                "refine_ms": round(ctx.kernel_ms("topk_refine"), 3),
                "topk_ms": round(ctx.kernel_ms("topk"), 3),
                "null_ratios_ms": round(ctx.kernel_ms("null_ratios"), 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
