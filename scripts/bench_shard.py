#!/usr/bin/env python3
"""Search time of ONE rank's row shard (what each GPU does in an N-GPU newref): rows [0, B/N) of the
15 kb problem on this device, for N = 1, 2, 4, 8 and candidate-segment counts (WCX_SCREEN_SEGMENTS)."""
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
    for n in (1, 2, 4, 8):
        rows = B // n
        d_idx = torch.empty((rows, 300), dtype=torch.int32, device=dev)
        d_dist = torch.empty((rows, 300), dtype=torch.float64, device=dev)
        d_nr = torch.empty((rows, len(ids)), dtype=torch.float64, device=dev)
        for seg in ("auto", "1", "2", "4"):
            if seg == "auto":
                os.environ.pop("WCX_SCREEN_SEGMENTS", None)
            else:
                os.environ["WCX_SCREEN_SEGMENTS"] = seg
            for _ in range(2):
                be.search(d_Xs, B, S, cum, 0, rows, 300, ids, d_idx, d_dist, d_nr)
                ctx.sync()
            out["N{}_seg{}".format(n, seg)] = {
                "screen_ms": round(ctx.kernel_ms("topk_screen"), 3),
                "refine_ms": round(ctx.kernel_ms("topk_refine"), 3),
                "topk_ms": round(ctx.kernel_ms("topk"), 3),
                "null_ratios_ms": round(ctx.kernel_ms("null_ratios"), 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
