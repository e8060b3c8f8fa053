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


def force_collectives():
    """WCX_FORCE_COLLECTIVES=1: the world == 1 short-circuits below are off -- a one-rank process group
    runs every collective of the multi-GPU path (all-gather of X and of the row blocks, the all-to-all
    of the hit records, the all-reduces of the sharded cut-off) through its real backend.  This is how
    RCCL is exercised on a one-GPU box (tests/test_gpu_rccl.py, bench.py's `rccl_world1`)."""
    import os
    if os.environ.get("WCX_FORCE_COLLECTIVES", "0") != "1":
        return False
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


COLLECTIVE_LOG = None      # a list: every collective of this module appends (op, bytes, backend, events)


def _logged(op, nbytes, cuda, fn, group=None):
    """Runs the collective `fn`; with COLLECTIVE_LOG set, its name, payload and -- on device tensors --
    a pair of events on the current stream around it (the process-group stream is joined to it by
    torch on both sides of a blocking collective) are recorded for collective_report()."""
    if COLLECTIVE_LOG is None:
        return fn()
    import torch
    import torch.distributed as dist
    ev = None
    if cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    import time
    t0 = time.perf_counter()
    out = fn()
    if ev:
        ev[1].record()
    COLLECTIVE_LOG.append({"op": op, "bytes": int(nbytes), "backend": dist.get_backend(group), "ev": ev,
                           "host_ms": 1e3 * (time.perf_counter() - t0)})
    return out


def collective_report():
    """COLLECTIVE_LOG with the event pairs resolved to milliseconds (synchronises); the log is emptied."""
    global COLLECTIVE_LOG
    out = []
    for e in COLLECTIVE_LOG or []:
        e = dict(e)
        ev = e.pop("ev")
        if ev:
            ev[1].synchronize()
            e["ms"] = ev[0].elapsed_time(ev[1])
        out.append(e)
    if COLLECTIVE_LOG is not None:
        COLLECTIVE_LOG = []
    return out


def _single(world):
    return world == 1 and not force_collectives()


def row_shard(rank, world, n_rows):
    """[begin, end) rows of `rank` -- the reference's part boundaries."""
    return _get_part(rank, world, n_rows)


def max_shard_rows(world, n_rows):
    return max(row_shard(r, world, n_rows)[1] - row_shard(r, world, n_rows)[0] for r in range(world))


def gather_padded(local_rows, world, group=None):
    """ONE collective: every rank's [pad, ...] buffer -> [world * pad, ...] on every rank (rank r's
    rows at r * pad; the rows beyond its shard are padding)."""
    import torch
    import torch.distributed as dist
    pad = local_rows.shape[0]
    local_rows = local_rows.contiguous()
    if dist.get_backend(group) == "gloo" and local_rows.is_cuda:
        # testing only (two ranks sharing one device): gloo gathers through host memory
        host = torch.empty((world * pad,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype)
        src = local_rows.cpu()
        _logged("all_gather_into_tensor", host.numel() * host.element_size(), False,
                lambda: dist.all_gather_into_tensor(host, src, group=group), group)
        return host.to(local_rows.device)
    gathered = torch.empty((world * pad,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype,
                           device=local_rows.device)
    _logged("all_gather_into_tensor", gathered.numel() * gathered.element_size(), gathered.is_cuda,
            lambda: dist.all_gather_into_tensor(gathered, local_rows, group=group), group)
    return gathered


def allgather_rows(local_rows, n_rows, world, group=None, backend=None):
    """local_rows: torch tensor [max_shard_rows, ...] whose first (end-begin) rows are this rank's
    shard (rest = padding).  Returns the dense [n_rows, ...] tensor on every rank.  With a GPU
    backend the padding is squeezed out by one device kernel (wcx_compact_rows_dev); the generic
    path (CPU tests) concatenates the shards."""
    import torch
    if _single(world):
        return local_rows[:n_rows]
    pad = local_rows.shape[0]
    gathered = gather_padded(local_rows, world, group)
    if backend is not None and hasattr(backend, "compact_rows") and gathered.is_cuda:
        return backend.compact_rows(gathered, world, pad, n_rows)
    parts = []
    for r in range(world):
        b, e = row_shard(r, world, n_rows)
        parts.append(gathered[r * pad: r * pad + (e - b)])
    return torch.cat(parts, 0)


class _GpuPredictMixin:
    """Row-sharded predict primitives on the GPU (device pointers)."""

    def wrap_rows(self, d_idx, d_dist, B, k, chr_cum, row0, nrows):
        from . import _lib
        cum, cum_p = _lib.i64_array(chr_cum)
        h = _lib.vp()
        _lib.check(self.ctx.lib.wcx_ref_wrap_rows_dev(self.ctx.h, d_idx.data_ptr(), d_dist.data_ptr(),
                                                      B, k, cum_p, len(cum), row0, nrows,
                                                      _lib.C.byref(h)))
        return h

    def free_ref(self, h):
        self.ctx.lib.wcx_ref_free(self.ctx.h, h)

    def moments(self, ref, cutoff, mean, phase):
        from . import _lib
        out = (_lib.C.c_double * 2)()
        _lib.check(self.ctx.lib.wcx_cutoff_moments_dev(self.ctx.h, ref, float(cutoff), float(mean),
                                                       int(phase), out))
        return out[0], out[1]

    def predict_pass(self, ref, x, cin, cout, cutoff, ct, build_mask, last, zB, rB, nB, lB):
        from . import _lib
        off = 8 * int(ct)      # outputs are written at (row - ct): shift so that they land at [row]
        _lib.check(self.ctx.lib.wcx_predict_pass_dev(
            self.ctx.h, ref, x.data_ptr(), cin.data_ptr(), cout.data_ptr(), float(cutoff), int(ct),
            int(build_mask), int(last), zB.data_ptr() + off, rB.data_ptr() + off,
            nB.data_ptr() + off, lB.data_ptr() + off))

    def nanmedian2(self, a0, a1):
        import torch
        from . import _lib
        out = torch.empty(2, dtype=torch.float64, device=a0.device)
        _lib.check(self.ctx.lib.wcx_nanmedian2_dev(self.ctx.h, a0.data_ptr(), a1.data_ptr(),
                                                   a0.numel(), out.data_ptr(), out.data_ptr() + 8))
        self.ctx.sync()
        o = out.cpu()
        return float(o[0]), float(o[1])


class GpuBackend(_GpuPredictMixin):
    # (see transpose() below: sample-major copy of the gathered row-major matrix)
    """Compute on this rank's MI355X through the C-ABI (device pointers).  The orchestration in this
    module interleaves torch ops (clones, all-gathers) with library calls on the same buffers, so
    the context MUST have been created on torch's current stream:
        _lib.Context(device, torch.cuda.current_stream().cuda_stream)"""

    def __init__(self, ctx):
        self.ctx = ctx

    def transpose(self, full):
        """[B][S] row-major -> [S][B] (tiled HIP kernel; torch's generic .t().contiguous() is 4x
        slower on this shape)."""
        import torch
        from . import _lib
        B, S = full.shape
        out = torch.empty((S, B), dtype=full.dtype, device=full.device)
        _lib.check(self.ctx.lib.wcx_transpose_dev(self.ctx.h, full.data_ptr(), B, S, out.data_ptr()))
        return out

    def gather_transpose(self, gathered, world, pad, B):
        """Padded row shards of X [world*pad][S] -> sample-major [S][B], padding skipped, in one
        kernel (no dense row-major intermediate)."""
        import torch
        from . import _lib
        S = gathered.shape[1]
        out = torch.empty((S, B), dtype=gathered.dtype, device=gathered.device)
        _lib.check(self.ctx.lib.wcx_gather_transpose_dev(self.ctx.h, gathered.data_ptr(), world, pad,
                                                         B, S, out.data_ptr()))
        return out

    def compact_rows(self, gathered, world, pad, B):
        """Padded row shards of a row-major table -> dense [B, ...] (one device copy kernel)."""
        import torch
        from . import _lib
        out = torch.empty((B,) + tuple(gathered.shape[1:]), dtype=gathered.dtype,
                          device=gathered.device)
        row_bytes = gathered[0].numel() * gathered.element_size()
        _lib.check(self.ctx.lib.wcx_compact_rows_dev(self.ctx.h, gathered.data_ptr(), world, pad, B,
                                                     row_bytes, out.data_ptr()))
        return out

    def search(self, d_Xs, B, S, chr_cum, row_begin, row_end, k, sample_ids, d_idx, d_dist, d_nr,
               mode=0):
        from . import _lib
        lib = self.ctx.lib
        cum, cum_p = _lib.i64_array(chr_cum)
        ids, ids_p = _lib.i32_array(sample_ids)
        # the ranking of the null samples needs X only: start it on the auxiliary stream now, it
        # runs beside the search
        # gonosomal pass (n_chr > 22): the autosomal target rows are dummies (index 0 / distance 1,
        # newref_tools.py:186-191); their null ratios are log2(x[row] / x[0]) without any gather
        ct = int(cum[21]) if len(cum) > 22 else 0
        real_lo = max(row_begin, ct)
        if real_lo < row_end:
            _lib.check(lib.wcx_null_rank_prepare_dev(self.ctx.h, d_Xs.data_ptr(), B, S, ids_p, len(ids)))
        _lib.check(lib.wcx_newref_topk_dev(self.ctx.h, d_Xs.data_ptr(), B, S, cum_p, len(cum),
                                           row_begin, row_end, k, mode, d_idx.data_ptr(),
                                           d_dist.data_ptr()))
        if row_begin < min(row_end, ct):
            _lib.check(lib.wcx_null_ratios_dummy_dev(self.ctx.h, d_Xs.data_ptr(), B, S, row_begin,
                                                     min(row_end, ct), ids_p, len(ids), d_nr.data_ptr()))
        if real_lo < row_end:
            skip = real_lo - row_begin
            _lib.check(lib.wcx_null_ratios_dev(self.ctx.h, d_Xs.data_ptr(), B, S,
                                               d_idx.data_ptr() + skip * k * 4, real_lo, row_end, k,
                                               ids_p, len(ids), d_nr.data_ptr() + skip * len(ids) * 8))

    # ---- the row-sharded symmetric sweep (wcx_newref_sym_*_dev): search() split around the exchange
    def sym_sweep(self, d_Xs, B, S, chr_cum, k, rank, world, bounds, sample_ids):
        """Phase 1: thresholds of all rows + this rank's tile pairs -> records.  Returns the number of
        records for every destination rank, or None where the library has no symmetric sweep for
        the shape (the caller then searches its row range the one-directional way)."""
        from . import _lib
        lib = self.ctx.lib
        cum, cum_p = _lib.i64_array(chr_cum)
        if len(cum) > 22 or world > 32:
            return None
        ids, ids_p = _lib.i32_array(sample_ids)
        bnd, bnd_p = _lib.i64_array(bounds)
        counts = (_lib.C.c_int64 * world)()
        if bounds[rank] < bounds[rank + 1]:
            _lib.check(lib.wcx_null_rank_prepare_dev(self.ctx.h, d_Xs.data_ptr(), B, S, ids_p, len(ids)))
        rc = lib.wcx_newref_sym_sweep_dev(self.ctx.h, d_Xs.data_ptr(), B, S, cum_p, len(cum), int(k), int(rank),
                                          int(world), bnd_p, counts)
        if rc == _lib.WCX_ERR_UNSUPPORTED:
            return None
        _lib.check(rc)
        return [int(c) for c in counts]

    def sym_records(self, send):
        from . import _lib
        _lib.check(self.ctx.lib.wcx_newref_sym_records_dev(self.ctx.h, send.data_ptr()))

    def sym_finish(self, recv, d_Xs, B, S, chr_cum, row_begin, row_end, k, sample_ids, d_idx, d_dist, d_nr):
        from . import _lib
        lib = self.ctx.lib
        ids, ids_p = _lib.i32_array(sample_ids)
        # recv None: the exchange was void (some rank's record pool overflowed): exact redo of the own rows
        n_recv = -1 if recv is None else int(recv.shape[0])
        _lib.check(lib.wcx_newref_sym_finish_dev(self.ctx.h, recv.data_ptr() if n_recv > 0 else None,
                                                 n_recv, d_idx.data_ptr(), d_dist.data_ptr()))
        if row_begin < row_end:
            _lib.check(lib.wcx_null_ratios_dev(self.ctx.h, d_Xs.data_ptr(), B, S, d_idx.data_ptr(), row_begin,
                                               row_end, k, ids_p, len(ids), d_nr.data_ptr()))


def _a2a_max_records():
    """Most records (16 bytes each) one rank hands ONE peer in one all_to_all_single call.  Measured on the
    MI355X box (scripts/debug_rccl_a2a.py, RCCL 2.26.6 under torch 2.10): a send / receive pair above 1 GiB
    delivers only its first half -- silently; up to 1 GiB everything arrives (all_gather_into_tensor has no
    such limit up to the 3 GiB tried).  The exchange therefore runs in rounds of at most 512 MiB per
    peer; 8 ranks at 15 kb x 500 need one (22 MB per peer), a one-rank group needs three."""
    import os
    return max(1, int(os.environ.get("WCX_A2A_MAX_RECORDS", str(32 << 20))))


def exchange_records(send, counts, world, group=None):
    """The ONE all-to-all of the row-sharded symmetric sweep: `send` holds this rank's records grouped by
    destination rank ([sum(counts), 4] int32, counts[r] of them for rank r); returns the records the
    other ranks (and this one) hold for THIS rank's rows, grouped by source rank.  The counts travel first
    (two integers per peer: the count for that peer and this rank's largest count, from which every rank
    derives the same number of ROUNDS -- see _a2a_max_records); a rank whose record pool overflowed sends
    -1 to everybody, which makes the exchange VOID on every rank alike (returns None: no further
    collective, each rank redoes its rows exactly)."""
    import torch
    import torch.distributed as dist
    cnts = [int(c) for c in counts]
    cmax = max(cnts) if cnts else 0
    cnt_in2 = torch.tensor([[c, cmax] for c in cnts], dtype=torch.int64)
    cnt_out2 = torch.empty((world, 2), dtype=torch.int64)
    host = dist.get_backend(group) == "gloo"
    if host:
        _logged("all_to_all_single", 16 * world, False, lambda: dist.all_to_all_single(cnt_out2, cnt_in2, group=group),
                group)
    else:
        ci, co = cnt_in2.to(send.device), cnt_out2.to(send.device)
        _logged("all_to_all_single", 16 * world, True, lambda: dist.all_to_all_single(co, ci, group=group), group)
        cnt_out2 = co.cpu()
    cnt_in, cnt_out = cnt_in2[:, 0], cnt_out2[:, 0]
    if bool((cnt_out < 0).any()) or bool((cnt_in < 0).any()):
        return None
    n_recv = int(cnt_out.sum())
    split_out, split_in = [int(c) for c in cnt_out], [int(c) for c in cnt_in]
    lim = _a2a_max_records()
    rounds = max(1, -(-int(cnt_out2[:, 1].max()) // lim))          # the same on every rank
    via_host = host and send.is_cuda       # testing only (ranks sharing one device): gloo exchanges through host memory
    src = send.cpu() if via_host else send
    recv = torch.empty((n_recv, 4), dtype=send.dtype, device=src.device)

    def a2a(dst, buf, so, si):
        _logged("all_to_all_single", 16 * int(buf.shape[0]), buf.is_cuda,
                lambda: dist.all_to_all_single(dst, buf, so, si, group=group), group)
    if rounds == 1:
        a2a(recv, src, split_out, split_in)
    else:
        off_in = [0] + list(np.cumsum(split_in))
        off_out = [0] + list(np.cumsum(split_out))
        for j in range(rounds):
            si = [min(max(c - j * lim, 0), lim) for c in split_in]
            so = [min(max(c - j * lim, 0), lim) for c in split_out]
            pieces = [src[off_in[r] + j * lim: off_in[r] + j * lim + si[r]] for r in range(world)]
            buf = pieces[0] if world == 1 else torch.cat(pieces, 0)
            got = torch.empty((sum(so), 4), dtype=send.dtype, device=src.device)
            a2a(got, buf.contiguous(), so, si)
            o = 0
            for r in range(world):
                recv[off_out[r] + j * lim: off_out[r] + j * lim + so[r]] = got[o: o + so[r]]
                o += so[r]
    return recv.to(send.device) if via_host else recv


def newref_sym_sharded(local_rows, n_rows, chr_cum, k, sample_ids, backend, rank, world, out=None):
    """Sharded build of an AUTOSOMAL pass with the symmetric sweep: newref_sharded's contract (same
    arguments, same results, bit for bit), but the tile PAIRS of the all-rows sweep are dealt out to the
    ranks -- each pair computed once, for both directions -- instead of every rank sweeping its target
    rows against all candidates.  Exchanges: the all-gather of X (as before) + ONE all-to-all of the hit
    records (16 bytes each, ~0.5 k per row at 15 kb).  Falls back to newref_sharded's search where the
    backend has no symmetric sweep for the shape (every rank takes the same branch: it depends on the
    shape alone)."""
    import torch
    if not _single(world) and hasattr(backend, "gather_transpose") and local_rows.is_cuda:
        Xs = backend.gather_transpose(gather_padded(local_rows, world), world, local_rows.shape[0],
                                      n_rows)
    else:
        full = allgather_rows(local_rows, n_rows, world).contiguous()
        Xs = backend.transpose(full) if hasattr(backend, "transpose") else full.t().contiguous()
    S = Xs.shape[0]
    b, e = row_shard(rank, world, n_rows)
    n = e - b
    dev = local_rows.device
    if out is None:
        out = (torch.empty((max(n, 1), k), dtype=torch.int32, device=dev),
               torch.empty((max(n, 1), k), dtype=torch.float64, device=dev),
               torch.empty((max(n, 1), len(sample_ids)), dtype=torch.float64, device=dev))
    bounds = [row_shard(r, world, n_rows)[0] for r in range(world)] + [n_rows]
    # Measured on one device (scripts/bench_shard_sym.py): a rank's wall at N = 2 / 4 / 8 ranks, symmetric
    # shard against one-directional shard of its row range --
    #   round 5 (profiles/r05/shard_sym_S500.json): 35.2 / 19.1 / 11.2 ms against 31.6 / 17.7 / 11.9 ms
    #   round 6 (profiles/r06/shard_sym_S500.json): 34.7 / 19.3 / 11.6 ms against 30.9 / 17.7 / 10.0 ms
    # -- every hit travels as a record here (dearer than a direct append) and prep + thresholds of ALL rows
    # (3 ms) do not shard; since the one-directional sweep takes its thresholds from hub counts (round 6:
    # its sweep of a 22.8 k-row shard 6.8 -> 5.0 ms) it wins at every N up to 8.  The symmetric form stays
    # available (WCX_SYM_SHARD_MIN = the rank count from which it is used; the tests and bench.py's one-rank
    # RCCL step set it) and is no longer a default below 16 ranks.
    import os
    sym_min = int(os.environ.get("WCX_SYM_SHARD_MIN", "16"))
    counts = backend.sym_sweep(Xs, n_rows, S, chr_cum, k, rank, world, bounds, sample_ids) \
        if world >= max(1 if force_collectives() else 2, sym_min) and hasattr(backend, "sym_sweep") else None
    if counts is None:
        backend.search(Xs, n_rows, S, chr_cum, b, e, k, sample_ids, out[0], out[1], out[2])
        return out[0][:n], out[1][:n], out[2][:n], Xs
    void = any(c < 0 for c in counts)
    send = torch.empty((0 if void else int(sum(counts)), 4), dtype=torch.int32, device=dev)
    if not void:
        backend.sym_records(send)
    recv = exchange_records(send, counts, world)
    backend.sym_finish(recv, Xs, n_rows, S, chr_cum, b, e, k, sample_ids, out[0], out[1], out[2])
    newref_sym_sharded.last_records = (max(0, int(sum(counts))), -1 if recv is None else int(recv.shape[0]))   # (bench / tests)
    return out[0][:n], out[1][:n], out[2][:n], Xs


def newref_sharded(local_rows, n_rows, chr_cum, k, sample_ids, backend, rank, world, out=None):
    """Sharded reference build on torch tensors.

    local_rows : [max_shard_rows, S] float64, this rank's rows of X (row-major), padded.
    Returns (idx [n_local,k] int32, dist [n_local,k] f64, nr [n_local,m] f64) for this rank's
    rows, plus the gathered sample-major matrix it was computed from."""
    import torch
    if not _single(world) and hasattr(backend, "gather_transpose") and local_rows.is_cuda:
        # the ONE exchange, then padded shards -> sample-major in one kernel
        Xs = backend.gather_transpose(gather_padded(local_rows, world), world, local_rows.shape[0],
                                      n_rows)
    else:
        full = allgather_rows(local_rows, n_rows, world).contiguous()
        Xs = backend.transpose(full) if hasattr(backend, "transpose") else full.t().contiguous()
    # ^ sample-major [S][B]
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


def newref_gonosomal_sharded(local_rows, n_rows, chr_cum, k, sample_ids, backend, rank, world, full):
    """A gonosomal pass (F: 23 chromosomes, M: 24) of the sharded reference build.  Only the X (and Y)
    rows are searched (newref_tools.py:186-191); with the reference's own _get_part over ALL rows
    they would sit in the last rank(s), so the GONOSOMAL rows [ct, n_rows) are what is split over
    the ranks here.  One all-gather of this pass's X, the search + null ratios of this rank's
    gonosomal rows, one all-gather of those (small) row blocks; the dummy autosomal rows (index 0,
    distance 1, null ratio log2(x / x[0])) are written by every rank itself.
    full = (idx [n_rows,k] int32, dist [n_rows,k] f64, nr [n_rows,m] f64) device buffers that
    receive the complete tables of the pass.  Returns them."""
    import torch
    if not _single(world) and hasattr(backend, "gather_transpose") and local_rows.is_cuda:
        Xs = backend.gather_transpose(gather_padded(local_rows, world), world, local_rows.shape[0], n_rows)
    else:
        fullX = allgather_rows(local_rows, n_rows, world).contiguous()
        Xs = backend.transpose(fullX) if hasattr(backend, "transpose") else fullX.t().contiguous()
    S = Xs.shape[0]
    ct = int(chr_cum[21])
    if _single(world):
        backend.search(Xs, n_rows, S, chr_cum, 0, n_rows, k, sample_ids, full[0], full[1], full[2])
        return full
    n_g = n_rows - ct
    a, b = _get_part(rank, world, n_g)
    pad = max_shard_rows(world, n_g)
    dev = local_rows.device
    loc = (torch.empty((pad, k), dtype=torch.int32, device=dev),
           torch.empty((pad, k), dtype=torch.float64, device=dev),
           torch.empty((pad, len(sample_ids)), dtype=torch.float64, device=dev))
    if b > a:
        backend.search(Xs, n_rows, S, chr_cum, ct + a, ct + b, k, sample_ids, loc[0], loc[1], loc[2])
    gathered = tuple(allgather_rows(t, n_g, world, backend=backend) for t in loc)
    if ct:
        backend.search(Xs, n_rows, S, chr_cum, 0, ct, k, sample_ids, full[0], full[1], full[2])   # dummies
    for dst, src in zip(full, gathered):
        dst[ct:n_rows].copy_(src)
    return full


def _padded(t, pad):
    import torch
    if t.shape[0] == pad:
        return t
    p = torch.zeros((pad,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    p[:t.shape[0]] = t
    return p


def gather_reference(idx_local, dist_local, n_rows, world, backend=None):
    """All-gather the finished row blocks so every rank (= predict replica) holds the whole
    reference.  Inputs are this rank's [n_local, k] blocks."""
    if _single(world):
        return idx_local, dist_local
    pad = max_shard_rows(world, n_rows)
    return (allgather_rows(_padded(idx_local, pad), n_rows, world, backend=backend),
            allgather_rows(_padded(dist_local, pad), n_rows, world, backend=backend))


def gather_reference3(idx_local, dist_local, nr_local, n_rows, world, backend=None):
    """gather_reference + the null-ratio rows: the three tables `newref` writes to disk."""
    if _single(world):
        return idx_local, dist_local, nr_local
    pad = max_shard_rows(world, n_rows)
    return tuple(allgather_rows(_padded(t, pad), n_rows, world, backend=backend)
                 for t in (idx_local, dist_local, nr_local))


def predict_one_dev(backend, idx, dist, nr, d_x, B, k, chr_cum, rem_input, pt):
    """predict of ONE sample against a reference resident on this device (main.py:191-279 for the
    autosomal pass): cut-off, weights, normalize_repeat, post-processing (minrefbins, inflation,
    log2 transform: wcx_post_process_dev) on the device; the three result vectors come back into
    pinned host buffers (they are what the tables are written from); CBS + segment z on the device
    with the null ratios inflated in place.
    Returns the reference's result rows [chr, start, end, z, ratio].  The per-chromosome arrays of
    the results live in buffers owned by `backend` and are valid until its next call."""
    import numpy as np
    import torch
    from . import _lib
    ctx = backend.ctx
    lib = ctx.lib
    dev = d_x.device
    cum, cum_p = _lib.i64_array(chr_cum)
    args = rem_input["args"]
    mask = np.asarray(rem_input["mask"], dtype=bool)
    n_bins = len(mask)
    cache = getattr(backend, "_predict_bufs", None)
    key = (B, n_bins, hash(mask.tobytes()))      # the mask's CONTENT: ids are reused after collection
    if cache is None or cache["key"] != key:
        cache = {"key": key,
                 "pos": torch.from_numpy(np.flatnonzero(mask).astype(np.int32)).to(dev),
                 "host": torch.empty((3, n_bins), dtype=torch.float64).pin_memory(),
                 "out": torch.empty((4, B), dtype=torch.float64, device=dev),
                 "med": torch.empty(2, dtype=torch.float64, device=dev)}
        if cache["pos"].numel() != B:
            raise ValueError("mask selects {} bins, the reference has {}".format(cache["pos"].numel(), B))
        backend._predict_bufs = cache
    out, med, host = cache["out"], cache["med"], cache["host"]
    h = _lib.vp()
    _lib.check(lib.wcx_ref_wrap_dev(ctx.h, idx.data_ptr(), dist.data_ptr(), B, k, cum_p, len(cum),
                                    _lib.C.byref(h)))
    try:
        cutoff = _lib.C.c_double()
        _lib.check(lib.wcx_cutoff(ctx.h, h, int(args.maskrepeats), _lib.C.byref(cutoff)))
        _lib.check(lib.wcx_weights_dev(ctx.h, h, out[3].data_ptr()))
        _lib.check(lib.wcx_predict_normalize_dev(ctx.h, h, d_x.data_ptr(), 1, cutoff.value, 0, 0,
                                                 out[0].data_ptr(), out[1].data_ptr(),
                                                 out[2].data_ptr(), med[0:].data_ptr(),
                                                 med[1:].data_ptr()))
        _lib.check(lib.wcx_post_process_dev(ctx.h, out[0].data_ptr(), out[1].data_ptr(),
                                            out[2].data_ptr(), out[3].data_ptr(), B,
                                            med[0:].data_ptr(), med[1:].data_ptr(),
                                            float(args.minrefbins), cache["pos"].data_ptr(), n_bins,
                                            host[0].data_ptr(), host[1].data_ptr(),
                                            host[2].data_ptr()))
    finally:
        lib.wcx_ref_free(ctx.h, h)
    off = np.concatenate(([0], np.cumsum(rem_input["bins_per_chr"]))).astype(int)
    n_chr = len(rem_input["bins_per_chr"])
    results = {}
    for row, key in enumerate(("results_r", "results_z", "results_w")):
        full = host[row].numpy()
        results[key] = [full[off[c]:off[c + 1]] for c in range(n_chr)]
    pt.attach_null_matrix_dev(nr, rem_input["mask"], ctx)
    results["results_nr"] = pt.ATTACHED
    return pt.exec_cbs(rem_input, results, ctx)


def predict_batch_dev(backend, A, G, d_xA, d_xG, rem_input, pt, want_host=False):
    """predict of a BATCH of samples, complete and device-resident (main.py:191-279): autosomal pass
    against A = {"idx", "dist", "nr", "cum"} (device tensors of the autosomal reference + its
    cumulative bin counts), gonosomal pass against G (the .F / .M reference: all its rows; None =
    autosomes only), cut-off on the AUTOSOMAL distances (predict_tools.py:75), the A + gonosome merge,
    minrefbins / inflation / log2 transform (wcx_post_process_merge_dev), CBS of all samples in one
    level-synchronous pass (wcx_cbs_batch_dev) and their segment z (wcx_segment_z_batch_dev) -- no
    NumPy between the normalisation and the segments.
    d_xA [ns][BA] / d_xG [ns][B of G]: the projected coverage vectors for the two references.
    rem_input["mask"] / ["bins_per_chr"] are those of the reference the gonosomes come from.
    Returns per sample the reference's result rows [chr, start, end, z, ratio] (+ the host copy of
    the per-bin r, z, w [3][ns][n_bins] if want_host)."""
    import numpy as np
    import torch
    from . import _lib
    ctx = backend.ctx
    lib = ctx.lib
    dev = d_xA.device
    args = rem_input["args"]
    mask = np.asarray(rem_input["mask"], dtype=bool)
    n_bins = len(mask)
    d_xA = d_xA.view(-1, d_xA.shape[-1])
    ns = int(d_xA.shape[0])
    BA, k = A["idx"].shape
    cumA, cumA_p = _lib.i64_array(A["cum"])
    ct = BG_all = BG = 0
    if G is not None:
        BG_all = G["idx"].shape[0]
        cumG, cumG_p = _lib.i64_array(G["cum"])
        ct = int(cumG[21])
        BG = BG_all - ct
        d_xG = d_xG.view(-1, d_xG.shape[-1])
    cache = getattr(backend, "_predict_full_bufs", None)
    key = (ns, BA, BG, BG_all, n_bins, hash(mask.tobytes()))
    if cache is None or cache["key"] != key:
        pos = np.flatnonzero(mask).astype(np.int32)
        if len(pos) != BA + BG:
            raise ValueError("mask selects {} bins, the references hold {} + {}".format(len(pos), BA, BG))
        cache = {"key": key, "pos": torch.from_numpy(pos).to(dev),
                 "a": torch.empty((3, ns, BA), dtype=torch.float64, device=dev),
                 "g": torch.empty((3, ns, max(BG, 1)), dtype=torch.float64, device=dev),
                 "wa": torch.empty(BA, dtype=torch.float64, device=dev),
                 "wg": torch.empty(max(BG_all, 1), dtype=torch.float64, device=dev),
                 "med": torch.empty((4, ns), dtype=torch.float64, device=dev),
                 "out": torch.empty((3, ns, n_bins), dtype=torch.float64, device=dev)}
        backend._predict_full_bufs = cache
    a, g, med, out = cache["a"], cache["g"], cache["med"], cache["out"]
    hA, hG = _lib.vp(), _lib.vp()
    w_fallback = _lib.C.c_int(0)
    _lib.check(lib.wcx_ref_wrap_dev(ctx.h, A["idx"].data_ptr(), A["dist"].data_ptr(), BA, k, cumA_p,
                                    len(cumA), _lib.C.byref(hA)))
    try:
        cutoff = _lib.C.c_double()
        _lib.check(lib.wcx_cutoff(ctx.h, hA, int(args.maskrepeats), _lib.C.byref(cutoff)))
        _lib.check(lib.wcx_weights_dev(ctx.h, hA, cache["wa"].data_ptr()))
        ctx.timer_tag("aut:")
        try:
            _lib.check(lib.wcx_predict_normalize_dev(ctx.h, hA, d_xA.data_ptr(), ns, cutoff.value, 0, 0,
                                                     a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(),
                                                     med[0].data_ptr(), med[1].data_ptr()))
        finally:
            ctx.timer_tag("")
        if G is not None:
            _lib.check(lib.wcx_ref_wrap_dev(ctx.h, G["idx"].data_ptr(), G["dist"].data_ptr(), BG_all, k,
                                            cumG_p, len(cumG), _lib.C.byref(hG)))
            ctx.timer_tag("gon:")             # (its own timer: "weights" alone would keep only this call)
            try:
                _lib.check(lib.wcx_weights_dev(ctx.h, hG, cache["wg"].data_ptr()))
            finally:
                ctx.timer_tag("")
            _lib.check(lib.wcx_predict_normalize_dev(ctx.h, hG, d_xG.data_ptr(), ns, cutoff.value, ct, 22,
                                                     g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(),
                                                     med[2].data_ptr(), med[3].data_ptr()))
        _lib.check(lib.wcx_post_process_merge_dev(
            ctx.h, a[0].data_ptr(), a[1].data_ptr(), a[2].data_ptr(), cache["wa"].data_ptr(), BA,
            g[0].data_ptr() if BG else None, g[1].data_ptr() if BG else None,
            g[2].data_ptr() if BG else None, cache["wg"].data_ptr() + 8 * ct if BG else None, BG, ns,
            med[0].data_ptr(), med[1].data_ptr(), float(args.minrefbins), cache["pos"].data_ptr(),
            n_bins, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), _lib.C.byref(w_fallback)))
    finally:
        lib.wcx_ref_free(ctx.h, hA)
        if hG:
            lib.wcx_ref_free(ctx.h, hG)
    if w_fallback.value:        # main.py:252-256
        import logging
        logging.warning("Non-numeric values found in weights -- reference too small. "
                        "Circular binary segmentation and z-scoring will be unweighted")
    if getattr(args, "blacklist", None):
        # main.py:263-265 / predict_tools.py:202-214: ratio, z and weight of the blacklisted bins
        # become 0 between the log2 transform and CBS (the merge call above has synchronised)
        bl = cache.get("blacklist")
        if bl is None or bl[0] != args.blacklist:
            bl = (args.blacklist, torch.from_numpy(pt.blacklist_bin_indices(rem_input)).to(dev))
            cache["blacklist"] = bl
        if bl[1].numel():       # (the context runs on torch's current stream: ordered with the kernels)
            out.index_fill_(2, bl[1], 0.0)
    # null ratios: autosomal rows + the gonosomal rows of the gonosomal reference (main.py:216-219)
    if G is None:
        nr = A["nr"]
    else:
        parts = [A["nr"], G["nr"][ct:]]
        m = max(p_.shape[1] for p_ in parts)          # ragged (fewer than 100 samples of one gender):
        parts = [p_ if p_.shape[1] == m else          # padded with NaN like main.py's host merge
                 torch.cat((p_, torch.full((p_.shape[0], m - p_.shape[1]), float("nan"), dtype=p_.dtype,
                                           device=p_.device)), 1) for p_ in parts]
        nr = torch.cat(parts, 0)
    pt.attach_null_matrix_dev(nr, rem_input["mask"], ctx)
    rows = pt.exec_cbs_batch_dev(rem_input, out[0], out[2], ctx)
    if want_host:
        return rows, out.cpu().numpy()
    return rows


def predict_full_dev(backend, A, G, d_xA, d_xG, rem_input, pt, want_host=False):
    """One sample of predict_batch_dev (d_xA [BA], d_xG [B of G])."""
    res = predict_batch_dev(backend, A, G, d_xA, d_xG, rem_input, pt, want_host)
    if want_host:
        return res[0][0], res[1][:, 0, :]
    return res[0]


def _allreduce2(a, b, world):
    if _single(world):
        return a, b
    import torch
    import torch.distributed as dist
    t = torch.tensor([a, b], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    _logged("all_reduce", 16, t.is_cuda, lambda: dist.all_reduce(t))
    return float(t[0]), float(t[1])


def cutoff_sharded(backend, ref, repeats, world):
    """get_optimal_cutoff (predict_tools.py:74-82) over row-sharded distances: per repeat two
    local moment sweeps + two tiny all-reduces."""
    cutoff = float("inf")
    for _ in range(repeats):
        s, c = _allreduce2(*backend.moments(ref, cutoff, 0.0, 0), world)
        mean = s / c
        ss, _ = _allreduce2(*backend.moments(ref, cutoff, mean, 1), world)
        cutoff = mean + 3.0 * (ss / c) ** 0.5
    return cutoff


def normalize_sharded(backend, ref, x, n_rows, ct, cutoff, rank, world):
    """normalize_repeat (predict_tools.py:94-108) of ONE sample with the reference rows sharded:
    each rank runs the three masked passes on its own rows; between passes the updated slices
    of test_copy are exchanged (all-gather of B doubles), at the end z / r / n / log2 r likewise.
    x: torch [B] float64 on the compute device.  Returns (z, r, n, m_lr, m_z) with z, r, n of
    length B - ct (full, identical on every rank)."""
    import torch
    B = n_rows
    b, e = row_shard(rank, world, B)
    pad = max_shard_rows(world, B)
    cin = x.clone()
    cout = x.clone()
    zB = torch.zeros(B, dtype=torch.float64, device=x.device)
    rB, nB, lB = torch.zeros_like(zB), torch.zeros_like(zB), torch.zeros_like(zB)

    def exchange(v):
        if _single(world):
            return v
        loc = torch.zeros(pad, dtype=v.dtype, device=v.device)
        loc[:e - b] = v[b:e]
        return allgather_rows(loc, B, world)

    for p in range(3):
        backend.predict_pass(ref, x, cin, cout, cutoff, ct, p == 0, p == 2, zB, rB, nB, lB)
        if p < 2:
            full = exchange(cout)
            cin, cout = full, full.clone()
    zB, rB, nB, lB = exchange(zB), exchange(rB), exchange(nB), exchange(lB)
    m_lr, m_z = backend.nanmedian2(lB[ct:].contiguous(), zB[ct:].contiguous())
    return zB[ct:], rB[ct:], nB[ct:], m_lr, m_z


def stripe(items, rank, world):
    """Samples of a predict batch handled by this rank (round-robin)."""
    return [x for i, x in enumerate(items) if i % world == rank]
