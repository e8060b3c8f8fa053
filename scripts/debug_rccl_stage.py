#!/usr/bin/env python3
"""dev: which stage of the forced-collective A pass (RCCL, one rank) changes the tables at 15 kb x 500."""
import os, sys, socket
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update({"WCX_FORCE_COLLECTIVES": "1", "WCX_SYM_SHARD_MIN": "1", "MASTER_ADDR": "127.0.0.1"})
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1])
import torch
import torch.distributed as dist
import bench
from wisecondorx_amd import _lib
from wisecondorx_amd import dist as wd
S = int(sys.argv[1]) if len(sys.argv) > 1 else 500
p = bench.make_full_workload(15000, S)[1]["A"]
X = p["X"]; cum = [int(v) for v in p["masked_bins_per_chr_cum"]]
B, k = cum[-1], 300
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
be = wd.GpuBackend(ctx)
Xrow = torch.from_numpy(np.ascontiguousarray(X)).to(dev)          # [B][S]
ids = np.ascontiguousarray(np.random.default_rng(5).permutation(S)[:100], dtype=np.int32)
Xs_ref = be.transpose(Xrow)
o0 = (torch.empty((B, k), dtype=torch.int32, device=dev), torch.empty((B, k), dtype=torch.float64, device=dev),
      torch.empty((B, len(ids)), dtype=torch.float64, device=dev))
be.search(Xs_ref, B, S, cum, 0, B, k, ids, *o0)
ctx.sync()
def cmp(tag, a, b):
    a8, b8 = a.contiguous().view(torch.uint8).reshape(a.shape[0], -1), b.contiguous().view(torch.uint8).reshape(b.shape[0], -1)
    rows = torch.nonzero((a8 != b8).any(dim=1)).flatten()
    print(tag, "rows differing", int(rows.numel()), rows[:6].tolist(), flush=True)
for rep in range(3):
    g = wd.gather_padded(Xrow, 1)
    cmp("rep %d all-gathered X" % rep, g, Xrow)
    Xs = be.gather_transpose(g, 1, Xrow.shape[0], B)
    cmp("rep %d gather_transpose" % rep, Xs, Xs_ref)
    bufs = tuple(torch.empty_like(t) for t in o0)
    i1, d1, n1, _ = wd.newref_sym_sharded(Xrow, B, cum, k, ids, be, 0, 1, out=bufs)
    ctx.sync()
    cmp("rep %d sym_sharded idx" % rep, i1, o0[0]); cmp("rep %d sym_sharded dist" % rep, d1, o0[1]); cmp("rep %d sym_sharded nr" % rep, n1, o0[2])
    fi, fd, fn = wd.gather_reference3(i1, d1, n1, B, 1, be)
    cmp("rep %d gathered idx" % rep, fi, o0[0]); cmp("rep %d gathered nr" % rep, fn, o0[2])
    # the exchange alone: records in, records out
    counts = be.sym_sweep(Xs_ref, B, S, cum, k, 0, 1, [0, B], ids)
    send = torch.empty((sum(counts), 4), dtype=torch.int32, device=dev)
    be.sym_records(send)
    recv = wd.exchange_records(send, counts, 1)
    cmp("rep %d records through all_to_all" % rep, recv, send)
    be.sym_finish(recv, Xs_ref, B, S, cum, 0, B, k, ids, *bufs)
    ctx.sync()
    cmp("rep %d finish after exchange idx" % rep, bufs[0], o0[0])
dist.destroy_process_group()
