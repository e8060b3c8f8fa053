"""CPU, world_size 2 and 8, gloo: the row-sharded reference build (partition + the one all-gather +
result gather) gives exactly the single-process result -- 8 uneven _get_part shards with padding
included.  The compute backend injected here is
the oracle (tests may use it); on the GPU the same orchestration drives libwcx_hip.so."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend:
    def search(self, Xs, B, S, chr_cum, row_begin, row_end, k, sample_ids, o_idx, o_dist, o_nr,
               mode=0):
        import torch
        from oracle import c_oracle as CO
        from oracle import wcx_oracle as O
        xs = np.ascontiguousarray(Xs.numpy())
        i, d = CO.get_reference_rows(xs, list(chr_cum), row_begin, row_end, k)
        nr = O.null_ratios(xs.T, i, row_begin, row_end, list(sample_ids))
        n = row_end - row_begin
        o_idx[:n] = torch.from_numpy(i)
        o_dist[:n] = torch.from_numpy(d)
        o_nr[:n] = torch.from_numpy(nr)


class OraclePredictBackend:
    """CPU stand-in for the row-sharded predict primitives (numpy oracle)."""

    def __init__(self, mbpc, cum):
        self.mbpc, self.cum = mbpc, cum

    def wrap_rows(self, idx, dist, B, k, chr_cum, row0, nrows):
        return {"idx": idx.numpy(), "dist": dist.numpy(), "row0": row0, "nrows": nrows, "B": B, "k": k}

    def moments(self, ref, cutoff, mean, phase):
        d = ref["dist"]
        sel = d[d < cutoff]
        if phase == 0:
            return float(sel.sum()), float(sel.size)
        return float(((sel - mean) ** 2).sum()), 0.0

    def predict_pass(self, ref, x, cin, cout, cutoff, ct, build_mask, last, zB, rB, nB, lB):
        import torch
        from oracle import wcx_oracle as O
        B, k, row0, nrows = ref["B"], ref["k"], ref["row0"], ref["nrows"]
        idx = np.zeros((B, k), dtype=np.int32)
        dist = np.full((B, k), 1e10)
        idx[row0:row0 + nrows] = ref["idx"]
        dist[row0:row0 + nrows] = ref["dist"]
        z, r, n = O.normalize_once(x.numpy(), cin.numpy().copy(), self.mbpc, self.cum, idx, dist,
                                   cutoff, ct, int(np.searchsorted(self.cum, ct, side="right")),
                                   row_range=(row0, row0 + nrows))
        lo, hi = max(ct, row0), row0 + nrows
        with np.errstate(all="ignore"):
            zB[lo:hi] = torch.from_numpy(z[lo - ct:hi - ct])
            rB[lo:hi] = torch.from_numpy(r[lo - ct:hi - ct])
            nB[lo:hi] = torch.from_numpy(n[lo - ct:hi - ct])
            lB[lo:hi] = torch.from_numpy(np.log2(r[lo - ct:hi - ct]))
            c = cin.numpy().copy()
            zz = z[lo - ct:hi - ct]
            c[lo:hi][np.abs(zz) >= O.Z_MASK] = -1
        cout[lo:hi] = torch.from_numpy(c[lo:hi])

    def nanmedian2(self, a0, a1):
        with np.errstate(all="ignore"):
            return float(np.nanmedian(a0.numpy())), float(np.nanmedian(a1.numpy()))


def _worker(rank, world, port, X, cum, k, ids, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from wisecondorx_amd import dist as wd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = X.shape[0]
    b, e = wd.row_shard(rank, world, B)
    pad = wd.max_shard_rows(world, B)
    local = torch.zeros((pad, X.shape[1]), dtype=torch.float64)
    local[:e - b] = torch.from_numpy(np.ascontiguousarray(X[b:e]))
    idx, dd, nr, Xs = wd.newref_sharded(local, B, cum, k, ids, OracleBackend(), rank, world)
    assert Xs.shape == (X.shape[1], B)
    fi, fd, fnr = wd.gather_reference3(idx, dd, nr, B, world)
    # row-sharded predict of one sample against the rows this rank just built
    mb = np.diff(np.concatenate(([0], cum))).tolist()
    pb = OraclePredictBackend(mb, list(cum))
    ref = pb.wrap_rows(idx, dd, B, k, cum, b, e - b)
    cutoff = wd.cutoff_sharded(pb, ref, 5, world)
    xt = torch.from_numpy(np.ascontiguousarray(X[:, 0]) * (1 + 0.2 * np.sin(np.arange(B))))
    z, r, n, mlr, mz = wd.normalize_sharded(pb, ref, xt, B, 0, cutoff, rank, world)
    q.put((rank, idx.numpy().copy(), dd.numpy().copy(), nr.numpy().copy(),
           fi.numpy().copy(), fd.numpy().copy(), fnr.numpy().copy(),
           (cutoff, z.numpy().copy(), r.numpy().copy(), n.numpy().copy(), mlr, mz)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_row_sharded_newref(world):
    import torch.multiprocessing as mp
    from oracle import wcx_oracle as O
    from wisecondorx_amd.synth import corrected_matrix
    X, mbpc, cum = corrected_matrix([40, 33, 27, 25, 20, 18], 12, seed=3)
    X = np.asfortranarray(X)
    k, ids = 15, [3, 1, 7, 0]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, X, cum, k, ids, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from wisecondorx_amd import dist as wd
    sizes = [wd.row_shard(r, world, X.shape[0])[1] - wd.row_shard(r, world, X.shape[0])[0] for r in range(world)]
    assert [len(r[1]) for r in res] == sizes and (world == 2 or len(set(sizes)) > 1)   # uneven at 8
    ei, ed, enr = O.get_reference(X, mbpc, cum, k, 1, 1, ids)
    assert np.array_equal(np.concatenate([r[1] for r in res]), ei)
    assert np.array_equal(np.concatenate([r[2] for r in res]), ed)
    assert np.array_equal(np.concatenate([r[3] for r in res]), enr)
    x = np.ascontiguousarray(X[:, 0]) * (1 + 0.2 * np.sin(np.arange(X.shape[0])))
    ecut = O.get_optimal_cutoff(ed, 5)
    ez, er, en, emlr, emz = O.normalize_repeat(x, mbpc, cum, ei, ed, ecut, 0, 0)
    for r in res:      # every replica holds the whole reference after the gather
        assert np.array_equal(r[4], ei) and np.array_equal(r[5], ed) and np.array_equal(r[6], enr)
        cutoff, z, rr, n, mlr, mz = r[7]
        np.testing.assert_allclose(cutoff, ecut, rtol=1e-12)
        np.testing.assert_allclose(z, ez, rtol=1e-9, atol=1e-12, equal_nan=True)
        np.testing.assert_allclose(rr, er, rtol=1e-12, equal_nan=True)
        assert np.array_equal(n, en)
        np.testing.assert_allclose([mlr, mz], [emlr, emz], rtol=1e-9, atol=1e-12)


class OracleSymBackend(OracleBackend):
    """CPU stand-in for the row-sharded SYMMETRIC sweep (dist.newref_sym_sharded): the unordered row
    pairs are dealt out to the ranks, each pair looked at once and turned into up to two records
    (row, partner) -- one per direction in which it lies within the row's threshold -- for the ranks
    that own the rows; sym_finish ranks a row's received partners exactly.  Thresholds = the exact
    k-th distance of every row (what the hub counts bound from above on the device)."""

    def sym_sweep(self, Xs, B, S, chr_cum, k, rank, world, bounds, sample_ids):
        from oracle import wcx_oracle as O
        X = np.ascontiguousarray(Xs.numpy().T)
        cum = np.asarray(chr_cum)
        chrom = np.searchsorted(cum, np.arange(B), side="right")
        D = np.stack([O.sq_distances(X, X[r]) for r in range(B)])
        D[chrom[:, None] == chrom[None, :]] = np.inf
        thr = np.sort(D, axis=1)[:, k - 1]
        recs = []
        for x in range(B):
            for y in range(x + 1, B):
                if (31 * x + y) % world != rank or chrom[x] == chrom[y]:
                    continue
                if D[x, y] <= thr[x]:
                    recs.append((x, y))
                if D[x, y] <= thr[y]:
                    recs.append((y, x))
        recs = np.array(recs, dtype=np.int64).reshape(-1, 2)
        owner = np.searchsorted(np.asarray(bounds[1:]), recs[:, 0], side="right")
        order = np.argsort(owner, kind="stable")
        self._send = recs[order]
        return [int(np.sum(owner == r)) for r in range(world)]

    def sym_records(self, send):
        import torch
        send[:, 0] = torch.from_numpy(self._send[:, 0].astype(np.int32))
        send[:, 1] = torch.from_numpy(self._send[:, 1].astype(np.int32))
        send[:, 2:] = 0

    def sym_finish(self, recv, Xs, B, S, chr_cum, row_begin, row_end, k, sample_ids, o_idx, o_dist, o_nr):
        import torch
        from oracle import wcx_oracle as O
        X = np.ascontiguousarray(Xs.numpy().T)
        cum = np.asarray(chr_cum)
        rec = recv.numpy()
        assert np.all((rec[:, 0] >= row_begin) & (rec[:, 0] < row_end)), "a record reached the wrong rank"
        idx = np.full((row_end - row_begin, k), -1, dtype=np.int32)
        dd = np.full((row_end - row_begin, k), 1e10)
        for r in range(row_begin, row_end):
            part = np.unique(rec[rec[:, 0] == r, 1])
            c = int(np.searchsorted(cum, r, side="right"))
            cs, ce = (int(cum[c - 1]) if c else 0), int(cum[c])
            d = O.sq_distances(X[part], X[r])
            ci = np.where(part < cs, part, part - (ce - cs))           # chromosome-excluded index space
            o = np.lexsort((ci, d))[:k]
            idx[r - row_begin, :len(o)] = ci[o]
            dd[r - row_begin, :len(o)] = d[o]
        nr = O.null_ratios(X, idx, row_begin, row_end, list(sample_ids))
        n = row_end - row_begin
        o_idx[:n] = torch.from_numpy(idx)
        o_dist[:n] = torch.from_numpy(dd)
        o_nr[:n] = torch.from_numpy(nr)


def _worker_sym(rank, world, port, X, cum, k, ids, q):
    sys.path.insert(0, ROOT)
    os.environ["WCX_SYM_SHARD_MIN"] = "2"
    if world == 8:
        os.environ["WCX_A2A_MAX_RECORDS"] = "7"      # the exchange in several rounds (dist._a2a_max_records)
    import torch
    import torch.distributed as dist
    from wisecondorx_amd import dist as wd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = X.shape[0]
    b, e = wd.row_shard(rank, world, B)
    pad = wd.max_shard_rows(world, B)
    local = torch.zeros((pad, X.shape[1]), dtype=torch.float64)
    local[:e - b] = torch.from_numpy(np.ascontiguousarray(X[b:e]))
    idx, dd, nr, Xs = wd.newref_sym_sharded(local, B, cum, k, ids, OracleSymBackend(), rank, world)
    q.put((rank, idx.numpy().copy(), dd.numpy().copy(), nr.numpy().copy(), wd.newref_sym_sharded.last_records))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_row_sharded_symmetric_newref(world):
    """dist.newref_sym_sharded over gloo: pairs dealt out to the ranks, the hit records routed to the
    rows' owners by one all-to-all (uneven counts, 8 uneven shards) -- every rank's row block equals
    the single-process oracle's."""
    import torch.multiprocessing as mp
    from oracle import wcx_oracle as O
    from wisecondorx_amd.synth import corrected_matrix
    X, mbpc, cum = corrected_matrix([40, 33, 27, 25, 20, 18], 12, seed=3)
    X = np.asfortranarray(X)
    k, ids = 15, [3, 1, 7, 0]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_sym, args=(r, world, port, X, cum, k, ids, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ei, ed, enr = O.get_reference(X, mbpc, cum, k, 1, 1, ids)
    assert np.array_equal(np.concatenate([r[1] for r in res]), ei)
    assert np.array_equal(np.concatenate([r[2] for r in res]), ed)
    assert np.array_equal(np.concatenate([r[3] for r in res]), enr)
    sent, got = sum(r[4][0] for r in res), sum(r[4][1] for r in res)
    assert sent == got and sent >= X.shape[0] * k            # every record arrived; at least k per row


def test_stripe_and_shards():
    from wisecondorx_amd import dist as wd
    assert wd.stripe(list(range(7)), 1, 3) == [1, 4]
    cover = []
    for r in range(8):
        b, e = wd.row_shard(r, 8, 182179)
        cover += [b, e]
    assert cover[0] == 0 and cover[-1] == 182179 and cover[1:-1:2] == cover[2::2]
