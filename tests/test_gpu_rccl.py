"""GPU: every collective of the multi-GPU path (dist.py) through RCCL itself -- backend "nccl", a process
group of ONE rank on cuda:0, the world == 1 short-circuits switched off (WCX_FORCE_COLLECTIVES=1):
  all_gather_into_tensor   X row shards (gather_padded + wcx_gather_transpose_dev), the finished row blocks
                           (gather_reference3 + wcx_compact_rows_dev), the gonosomal row blocks, the slices
                           of the row-sharded normalisation
  all_to_all_single        counts + hit records of the row-sharded symmetric sweep (uneven split lists)
  all_reduce               the moments of the sharded cut-off
Results against the oracle, like the gloo tests of test_gpu_dist2.py -- what is new here is the backend:
device tensors, RCCL's stream ordering against the library's kernels on torch's current stream, dtypes.
(The reference's only parallel seam is newref_control.py:90-109 / newref_tools.py:244-247; no scaling is
measured here -- one rank.)"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init(port):
    sys.path.insert(0, ROOT)
    os.environ["WCX_FORCE_COLLECTIVES"] = "1"
    os.environ["WCX_SYM_SHARD_MIN"] = "1"
    os.environ["WCX_A2A_MAX_RECORDS"] = "400000"      # the record exchange of the test problem in several rounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    from wisecondorx_amd import _lib
    from wisecondorx_amd import dist as wd
    assert wd.force_collectives()
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    return torch, dist, dev, ctx, wd, wd.GpuBackend(ctx)


def _worker_autosomal(port, X, cum, k, ids, q):
    try:
        torch, dist, dev, ctx, wd, be = _init(port)
        B = X.shape[0]
        local = torch.from_numpy(np.ascontiguousarray(X)).to(dev)
        wd.COLLECTIVE_LOG = []
        # (1) row shards: ONE all-gather of X, one-directional search of "this rank's" rows
        idx, dd, nr, Xs = wd.newref_sharded(local, B, cum, k, ids, be, 0, 1)
        ctx.sync()
        r1 = (idx.cpu().numpy().copy(), dd.cpu().numpy().copy(), nr.cpu().numpy().copy())
        # (2) the symmetric sweep's tile pairs -> records -> ONE all-to-all (+ its counts)
        wd.newref_sym_sharded.last_records = None
        idx2, dd2, nr2, _ = wd.newref_sym_sharded(local, B, cum, k, ids, be, 0, 1)
        ctx.sync()
        st = ctx.topk_stats()
        r2 = (idx2.cpu().numpy().copy(), dd2.cpu().numpy().copy(), nr2.cpu().numpy().copy(),
              wd.newref_sym_sharded.last_records, st["fallback_rows"])
        # (3) the finished row blocks: padded all-gather + compaction
        fi, fd, fnr = wd.gather_reference3(idx2, dd2, nr2, B, 1, be)
        r3 = (fi.cpu().numpy().copy(), fd.cpu().numpy().copy(), fnr.cpu().numpy().copy())
        # (4) row-sharded predict: all-reduced moments, all-gathered slices between the passes
        h = be.wrap_rows(idx2, dd2, B, k, cum, 0, B)
        cutoff = wd.cutoff_sharded(be, h, 5, 1)
        xt = torch.from_numpy(np.ascontiguousarray(X[:, 0]) * (1 + 0.2 * np.sin(np.arange(B)))).to(dev)
        z, r, n, mlr, mz = wd.normalize_sharded(be, h, xt, B, 0, cutoff, 0, 1)
        ctx.sync()
        be.free_ref(h)
        r4 = (cutoff, z.cpu().numpy().copy(), r.cpu().numpy().copy(), n.cpu().numpy().copy(), mlr, mz)
        log = wd.collective_report()
        # A record exchange beyond 1 GiB: RCCL 2.26.6 delivers only the first half of a send / receive
        # pair above 1 GiB (scripts/debug_rccl_a2a.py) -- exchange_records must split it into rounds
        os.environ["WCX_A2A_MAX_RECORDS"] = str(32 << 20)           # the production value: 512 MiB
        n_big = 80_000_000                                            # 1.28 GB
        big = torch.arange(n_big * 4, dtype=torch.int32, device=dev).reshape(n_big, 4)
        got = wd.exchange_records(big, [n_big], 1)
        torch.cuda.synchronize()
        big_ok = bool(torch.equal(got, big))
        del big, got
        dist.barrier()
        dist.destroy_process_group()
        q.put(("ok", r1, r2, r3, r4, log, big_ok))
    except Exception as e:            # (the parent must not wait for its timeout)
        import traceback
        q.put(("error", "{}\n{}".format(e, traceback.format_exc())))


def _spawn(target, args):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    p = mpc.Process(target=target, args=(port,) + tuple(args) + (q,))
    p.start()
    res = q.get(timeout=900)
    p.join(timeout=120)
    assert res[0] == "ok", res[1]
    assert p.exitcode == 0
    return res[1:]


def test_rccl_world1_autosomal_build_and_sharded_predict():
    from oracle import c_oracle as CO
    from oracle import wcx_oracle as O
    from wisecondorx_amd.synth import corrected_matrix
    X, mbpc, cum = corrected_matrix([7100, 6600, 6100, 5400, 4700, 3901], 256, seed=29)    # 33 801 rows, K = 272
    X = np.asfortranarray(X)
    B, k, ids = cum[-1], 64, [3, 1, 7, 0, 22, 39, 255, 100]
    r1, r2, r3, r4, log, big_ok = _spawn(_worker_autosomal, (X, cum, k, ids))
    assert big_ok, "a 1.28 GB record exchange did not arrive intact"
    ei, ed = CO.get_reference_rows_threaded(np.ascontiguousarray(X.T), cum, 0, B, k)
    for tag, (gi, gd) in (("row shards", r1[:2]), ("symmetric shards", r2[:2]), ("gathered tables", r3[:2])):
        bad = np.flatnonzero((gi != ei).any(axis=1) | (gd != ed).any(axis=1))
        assert bad.size == 0, "{}: {} of {} rows differ (first {})".format(tag, bad.size, B, bad[:5])
    assert r2[3] is not None and r2[3][0] > 0 and r2[3][0] == r2[3][1], "the all-to-all of the records did not run"
    assert r2[4] <= 64
    for lo in (0, B // 2 - 500, B - 1000):
        with np.errstate(all="ignore"):
            enr = O.null_ratios(X, ei[lo:lo + 1000], lo, lo + 1000, ids)
        for gnr in (r1[2], r2[2], r3[2]):
            np.testing.assert_allclose(gnr[lo:lo + 1000], enr, rtol=1e-12, atol=1e-13)
    x = np.ascontiguousarray(X[:, 0]) * (1 + 0.2 * np.sin(np.arange(B)))
    ecut = O.get_optimal_cutoff(ed, 5)
    ez, er, en, emlr, emz = O.normalize_repeat(x, mbpc, cum, ei, ed, ecut, 0, 0)
    cutoff, z, rr, n, mlr, mz = r4
    np.testing.assert_allclose(cutoff, ecut, rtol=1e-12)
    np.testing.assert_allclose(z, ez, rtol=1e-9, atol=1e-12, equal_nan=True)
    np.testing.assert_allclose(rr, er, rtol=1e-12, equal_nan=True)
    assert np.array_equal(n, en)
    np.testing.assert_allclose([mlr, mz], [emlr, emz], rtol=1e-9, atol=1e-12)
    # which collectives RCCL executed (name -> calls): all three kinds must be there
    kinds = {e["op"] for e in log}
    assert {"all_gather_into_tensor", "all_to_all_single", "all_reduce"} <= kinds, log
    assert sum(1 for e in log if e["op"] == "all_to_all_single") >= 3        # counts + more than one round
    assert all(e["backend"] == "nccl" for e in log)


def _worker_gonosomal(port, X, cum, k, ids, xA, xG, XA, cumA, rem, q):
    try:
        torch, dist, dev, ctx, wd, be = _init(port)
        from wisecondorx_amd import predict_tools as pt
        B, BA = X.shape[0], XA.shape[0]
        wd.COLLECTIVE_LOG = []
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        m = len(ids)
        full = (torch.empty((B, k), dtype=torch.int32, device=dev), torch.empty((B, k), dtype=torch.float64, device=dev),
                torch.empty((B, m), dtype=torch.float64, device=dev))
        # gonosomal pass: all-gather of X, search of the gonosomal rows, all-gather of their row blocks
        gi, gd, gnr = wd.newref_gonosomal_sharded(t(X), B, cum, k, ids, be, 0, 1, full)
        # autosomal pass + gathered tables (every predict replica holds the whole reference)
        ai, ad, anr, _ = wd.newref_sharded(t(XA), BA, cumA, k, ids, be, 0, 1)
        ai, ad, anr = wd.gather_reference3(ai, ad, anr, BA, 1, be)
        ctx.sync()
        A = {"idx": ai, "dist": ad, "nr": anr, "cum": np.asarray(cumA, dtype=np.int64)}
        G = {"idx": gi, "dist": gd, "nr": gnr, "cum": np.asarray(cum, dtype=np.int64)}
        # a batch of samples against the RCCL-gathered tables, device-resident end to end
        rows, host = wd.predict_batch_dev(be, A, G, t(xA), t(xG), rem, pt, want_host=True)
        ctx.sync()
        out = (gi.cpu().numpy().copy(), gd.cpu().numpy().copy(), gnr.cpu().numpy().copy(),
               ai.cpu().numpy().copy(), ad.cpu().numpy().copy(), anr.cpu().numpy().copy(), rows, host)
        log = wd.collective_report()
        dist.barrier()
        dist.destroy_process_group()
        q.put(("ok", out, log))
    except Exception as e:
        import traceback
        q.put(("error", "{}\n{}".format(e, traceback.format_exc())))


def test_rccl_world1_gonosomal_pass_and_batch_predict():
    """newref_gonosomal_sharded and predict_batch_dev behind RCCL-gathered tables; the batch's rows and
    per-bin vectors equal those of the same calls on tables built WITHOUT any collective (one-process
    path, same device), which the other GPU tests pin to the oracle."""
    import argparse
    import torch
    from oracle import c_oracle as CO
    from wisecondorx_amd import _lib, newref_tools, predict_tools as pt
    from wisecondorx_amd import dist as wd
    from wisecondorx_amd.synth import corrected_matrix
    mb = [700, 660, 620, 580, 540, 500, 470, 440, 410, 380, 350, 330, 310, 290, 270, 250, 230, 210, 190, 170,
          150, 140, 420, 60]
    X, mbpc, cum = corrected_matrix(mb, 48, seed=5)
    X = np.asfortranarray(X)
    B, k, ids = cum[-1], 50, [5, 2, 40, 17, 0, 33]
    ct = cum[21]
    XA, cumA = np.asfortranarray(X[:ct]), list(cum[:22])
    rng = np.random.default_rng(8)
    ns = 5
    xG = np.ascontiguousarray(X[:, :ns].T) * (1 + 0.05 * rng.standard_normal((ns, B)))
    xG[1, 3000:3040] *= 1.4
    xA = np.ascontiguousarray(xG[:, :ct])
    # the unmasked layout: every chromosome has two bins more, masked out at random positions
    bpc = np.asarray(mb) + 2
    mask = np.ones(int(bpc.sum()), dtype=bool)
    o = 0
    for c in range(24):
        mask[o + rng.choice(bpc[c], 2, replace=False)] = False
        o += int(bpc[c])
    args = argparse.Namespace(minrefbins=10, maskrepeats=3, alpha=1e-3, seed=3)
    rem = {"args": args, "mask": mask, "bins_per_chr": bpc, "binsize": 100000, "ref_gender": "M"}
    out, log = _spawn(_worker_gonosomal, (X, cum, k, ids, xA, xG, XA, cumA, rem))
    gi, gd, gnr, ai, ad, anr, rows, host = out
    # the gonosomal rows against the oracle; the autosomal rows of the pass are dummies (newref_tools.py:186-191)
    ei, ed = CO.get_reference_rows(np.ascontiguousarray(X.T), cum, ct, B, k)
    assert np.array_equal(gi[ct:], ei) and np.array_equal(gd[ct:], ed)
    assert np.all(gi[:ct] == 0) and np.all(gd[:ct] == 1.0)
    eai, ead = CO.get_reference_rows(np.ascontiguousarray(XA.T), cumA, 0, ct, k)
    assert np.array_equal(ai, eai) and np.array_equal(ad, ead)
    # the same batch on tables uploaded from the host (no collective anywhere)
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    be = wd.GpuBackend(ctx)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    A = {"idx": t(ai), "dist": t(ad), "nr": t(anr), "cum": np.asarray(cumA, dtype=np.int64)}
    G = {"idx": t(gi), "dist": t(gd), "nr": t(gnr), "cum": np.asarray(cum, dtype=np.int64)}
    rows0, host0 = wd.predict_batch_dev(be, A, G, t(xA), t(xG), rem, pt, want_host=True)
    assert rows == rows0
    assert np.array_equal(host, host0, equal_nan=True)
    assert sum(len(r_) for r_ in rows) >= 24 * ns
    kinds = [e["op"] for e in log]
    assert kinds.count("all_gather_into_tensor") >= 2 + 3 + 3      # X twice, 3 gonosomal blocks, 3 tables
    assert all(e["backend"] == "nccl" for e in log)
