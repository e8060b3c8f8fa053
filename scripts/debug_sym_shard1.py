#!/usr/bin/env python3
"""dev: the row-sharded symmetric sweep with ONE part against the unsharded search and the oracle (15 kb x S)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from oracle import c_oracle as CO
from wisecondorx_amd import _lib
from wisecondorx_amd import dist as wd
S = int(sys.argv[1]) if len(sys.argv) > 1 else 500
p = bench.make_full_workload(15000, S)[1]["A"]
X = p["X"]; cum = [int(v) for v in p["masked_bins_per_chr_cum"]]
B, k = cum[-1], 300
dev = torch.device("cuda", 0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
be = wd.GpuBackend(ctx)
Xs = torch.from_numpy(np.ascontiguousarray(np.asarray(X).T)).to(dev)
ids = list(range(8))
o0 = (torch.empty((B, k), dtype=torch.int32, device=dev), torch.empty((B, k), dtype=torch.float64, device=dev),
      torch.empty((B, len(ids)), dtype=torch.float64, device=dev))
be.search(Xs, B, S, cum, 0, B, k, ids, *o0)
ctx.sync()
st0 = ctx.topk_stats()
o1 = tuple(torch.empty_like(t) for t in o0)
for rep in range(2):
    counts = be.sym_sweep(Xs, B, S, cum, k, 0, 1, [0, B], ids)
    send = torch.empty((sum(counts), 4), dtype=torch.int32, device=dev)
    be.sym_records(send)
    be.sym_finish(send, Xs, B, S, cum, 0, B, k, ids, *o1)
    ctx.sync()
    st1 = ctx.topk_stats()
    i0, d0, i1, d1 = o0[0].cpu().numpy(), o0[1].cpu().numpy(), o1[0].cpu().numpy(), o1[1].cpu().numpy()
    bad = np.flatnonzero((i0 != i1).any(axis=1) | (d0 != d1).any(axis=1))
    print("rep", rep, "records", counts, "rows differing", bad.size, bad[:10], "fallback", st0["fallback_rows"], st1["fallback_rows"])
if bad.size:
    rows = bad[:6]
    Xh = np.ascontiguousarray(np.asarray(X).T)
    for t in rows:
        oi, od = CO.get_reference_rows(Xh, cum, int(t), int(t) + 1, k)
        print("row", int(t), "unsharded == oracle", bool(np.array_equal(i0[t], oi[0]) and np.array_equal(d0[t], od[0])),
              "sharded == oracle", bool(np.array_equal(i1[t], oi[0]) and np.array_equal(d1[t], od[0])),
              "first diff col", int(np.flatnonzero(i1[t] != oi[0])[0]) if (i1[t] != oi[0]).any() else -1,
              "n wrong", int((i1[t] != oi[0]).sum()))
