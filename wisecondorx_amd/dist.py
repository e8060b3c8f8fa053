"""Multi-GPU plumbing: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

newref shards by target rows with the reference's own part formula
(newref_tools._get_part, newref_tools.py:244-247).  Each rank holds the rows of the bin-feature
matrix it prepared; every rank needs every candidate row, so there is exactly ONE exchange per
reference build: an all-gather of the row shards.  Results are disjoint row blocks.
predict runs as replicas: the finished reference is all-gathered once and samples are striped
over the ranks -- no collective on the per-sample path.

torch is plumbing here (device memory, streams, collectives); the compute is injected as a
`backend` object so the same orchestration runs on the GPU library and, in the CPU tests, on a
CPU stand-in supplied by the test.
"""
import numpy as np

from .newref_tools import _get_part


def row_shard(rank, world, n_rows):
    """[begin, end) rows of `rank` -- the reference's part boundaries."""
    return _get_part(rank, world, n_rows)


def max_shard_rows(world, n_rows):
    return max(row_shard(r, world, n_rows)[1] - row_shard(r, world, n_rows)[0] for r in range(world))


def allgather_rows(local_rows, n_rows, world, group=None):
    """local_rows: torch tensor [max_shard_rows, ...] whose first (end-begin) rows are this rank's
    shard (rest = padding).  Returns the full [n_rows, ...] tensor on every rank."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local_rows[:n_rows]
    pad = local_rows.shape[0]
    gathered = torch.empty((world * pad,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype,
                           device=local_rows.device)
    dist.all_gather_into_tensor(gathered, local_rows.contiguous(), group=group)
    parts = []
    for r in range(world):
        b, e = row_shard(r, world, n_rows)
        parts.append(gathered[r * pad: r * pad + (e - b)])
    return torch.cat(parts, 0)


class GpuBackend:
    """Compute on this rank's MI355X through the C-ABI (device pointers, caller's stream)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def search(self, d_Xs, B, S, chr_cum, row_begin, row_end, k, sample_ids, d_idx, d_dist, d_nr,
               mode=0):
        from . import _lib
        lib = self.ctx.lib
        cum, cum_p = _lib.i64_array(chr_cum)
        ids, ids_p = _lib.i32_array(sample_ids)
        _lib.check(lib.wcx_newref_topk_dev(self.ctx.h, d_Xs.data_ptr(), B, S, cum_p, len(cum),
                                           row_begin, row_end, k, mode, d_idx.data_ptr(),
                                           d_dist.data_ptr()))
        _lib.check(lib.wcx_null_ratios_dev(self.ctx.h, d_Xs.data_ptr(), B, S, d_idx.data_ptr(),
                                           row_begin, row_end, k, ids_p, len(ids),
                                           d_nr.data_ptr()))


def newref_sharded(local_rows, n_rows, chr_cum, k, sample_ids, backend, rank, world, out=None):
    """Sharded reference build on torch tensors.

    local_rows : [max_shard_rows, S] float64, this rank's rows of X (row-major), padded.
    Returns (idx [n_local,k] int32, dist [n_local,k] f64, nr [n_local,m] f64) for this rank's
    rows, plus the gathered sample-major matrix it was computed from."""
    import torch
    full = allgather_rows(local_rows, n_rows, world)            # the ONE exchange
    Xs = full.t().contiguous()                                   # sample-major [S][B]
    S = Xs.shape[0]
    b, e = row_shard(rank, world, n_rows)
    n = e - b
    dev = local_rows.device
    if out is None:
        out = (torch.empty((max(n, 1), k), dtype=torch.int32, device=dev),
               torch.empty((max(n, 1), k), dtype=torch.float64, device=dev),
               torch.empty((max(n, 1), len(sample_ids)), dtype=torch.float64, device=dev))
    backend.search(Xs, n_rows, S, chr_cum, b, e, k, sample_ids, out[0], out[1], out[2])
    return out[0][:n], out[1][:n], out[2][:n], Xs


def gather_reference(idx_local, dist_local, n_rows, world):
    """All-gather the finished row blocks so every rank (= predict replica) holds the whole
    reference.  Inputs are this rank's [n_local, k] blocks."""
    import torch
    if world == 1:
        return idx_local, dist_local
    pad = max_shard_rows(world, n_rows)

    def padded(t):
        p = torch.zeros((pad,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        p[:t.shape[0]] = t
        return p
    return (allgather_rows(padded(idx_local), n_rows, world),
            allgather_rows(padded(dist_local), n_rows, world))


def stripe(items, rank, world):
    """Samples of a predict batch handled by this rank (round-robin)."""
    return [x for i, x in enumerate(items) if i % world == rank]
