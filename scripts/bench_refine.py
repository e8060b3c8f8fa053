#!/usr/bin/env python3
"""dev helper (GPU box): refine time of the 15 kb search vs the sample-chunk size of the refine
(WCX_REFINE_CHUNK), results checked identical to the single-pass refine.  usage: bench_refine.py S [chunks...]"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from wisecondorx_amd import _lib
    from wisecondorx_amd import dist as wd
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    chunks = [int(a) for a in sys.argv[2:]] or [0]
    co, p, _ = bench.make_workload(15000, S)
    X = p["X"]
    cum = np.asarray(p["masked_bins_per_chr_cum"], dtype=np.int64)
    B = int(cum[-1])
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    be = wd.GpuBackend(ctx)
    d_Xs = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)
    ids = np.arange(min(S, 100), dtype=np.int32)
    d_idx = torch.empty((B, 300), dtype=torch.int32, device=dev)
    d_dist = torch.empty((B, 300), dtype=torch.float64, device=dev)
    d_nr = torch.empty((B, len(ids)), dtype=torch.float64, device=dev)
    ref = None
    for ch in chunks:
        os.environ["WCX_REFINE_CHUNK"] = str(ch)
        ms = []
        for _ in range(6):
            be.search(d_Xs, B, S, cum, 0, B, 300, ids, d_idx, d_dist, d_nr)
            ctx.sync()
            ms.append((ctx.kernel_ms("topk_refine"), ctx.kernel_ms("topk_screen"), ctx.kernel_ms("topk")))
        h = hashlib.sha256(d_idx.cpu().numpy().tobytes() + d_dist.cpu().numpy().tobytes()).hexdigest()[:16]
        ref = ref or h
        print(json.dumps({"S": S, "chunk": ch, "refine_ms": round(min(m[0] for m in ms[2:]), 3),
                          "screen_ms": round(min(m[1] for m in ms[2:]), 3), "topk_ms": round(min(m[2] for m in ms[2:]), 3),
                          "same_result": h == ref, "refined": ctx.topk_stats()["refined"]}), flush=True)


if __name__ == "__main__":
    main()
