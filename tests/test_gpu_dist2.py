"""GPU, world_size 2 and 8 over gloo with ALL ranks on cuda:0: the multi-GPU orchestration of dist.py
(row shards, the all-gather of X, sharded cut-off and normalisation) driving the real HIP library
(GpuBackend), compared with the oracle.  RCCL itself needs one device per rank and is exercised
by bench.py on the multi-GPU node; everything around the collectives is covered here."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, X, cum, k, ids, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from wisecondorx_amd import _lib
    from wisecondorx_amd import dist as wd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    be = wd.GpuBackend(ctx)
    B = X.shape[0]
    b, e = wd.row_shard(rank, world, B)
    pad = wd.max_shard_rows(world, B)
    local = torch.zeros((pad, X.shape[1]), dtype=torch.float64, device=dev)
    local[:e - b] = torch.from_numpy(np.ascontiguousarray(X[b:e])).to(dev)
    idx, dd, nr, Xs = wd.newref_sharded(local, B, cum, k, ids, be, rank, world)
    fi, fd, fnr = wd.gather_reference3(idx, dd, nr, B, world, be)     # padded all-gather + compact_rows
    h = be.wrap_rows(idx, dd, B, k, cum, b, e - b)
    cutoff = wd.cutoff_sharded(be, h, 5, world)
    xt = torch.from_numpy(np.ascontiguousarray(X[:, 0]) * (1 + 0.2 * np.sin(np.arange(B)))).to(dev)
    z, r, n, mlr, mz = wd.normalize_sharded(be, h, xt, B, 0, cutoff, rank, world)
    ctx.sync()
    be.free_ref(h)
    import hashlib
    full_sha = hashlib.sha256(fi.cpu().numpy().tobytes() + fd.cpu().numpy().tobytes() +
                              fnr.cpu().numpy().tobytes()).hexdigest()
    q.put((rank, idx.cpu().numpy().copy(), dd.cpu().numpy().copy(), nr.cpu().numpy().copy(),
           (cutoff, z.cpu().numpy().copy(), r.cpu().numpy().copy(), n.cpu().numpy().copy(), mlr, mz), full_sha))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_on_one_device_real_kernels(world):
    import torch.multiprocessing as mp
    from oracle import c_oracle as CO
    from oracle import wcx_oracle as O
    from wisecondorx_amd.synth import corrected_matrix
    X, mbpc, cum = corrected_matrix([900, 800, 700, 600, 503], 40, seed=23)      # 3503 rows: uneven in 8
    X = np.asfortranarray(X)
    B, k, ids = cum[-1], 64, [3, 1, 7, 0, 22, 39]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, world, port, X, cum, k, ids, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ei, ed = CO.get_reference_rows(np.ascontiguousarray(X.T), cum, 0, B, k)
    assert np.array_equal(np.concatenate([r[1] for r in res]), ei)
    assert np.array_equal(np.concatenate([r[2] for r in res]), ed)
    with np.errstate(all="ignore"):
        enr = O.null_ratios(X, ei, 0, B, ids)
    np.testing.assert_allclose(np.concatenate([r[3] for r in res]), enr, rtol=1e-12, atol=1e-13)
    # every rank's gathered copy of the finished reference (idx | dist | null ratios) is the same
    # dense table = the concatenation of the shards
    import hashlib
    want = hashlib.sha256(np.concatenate([r[1] for r in res]).tobytes() + np.concatenate([r[2] for r in res]).tobytes()
                          + np.concatenate([r[3] for r in res]).tobytes()).hexdigest()
    assert all(r[5] == want for r in res)
    x = np.ascontiguousarray(X[:, 0]) * (1 + 0.2 * np.sin(np.arange(B)))
    ecut = O.get_optimal_cutoff(ed, 5)
    ez, er, en, emlr, emz = O.normalize_repeat(x, mbpc, cum, ei, ed, ecut, 0, 0)
    for r in res:
        cutoff, z, rr, n, mlr, mz = r[4]
        np.testing.assert_allclose(cutoff, ecut, rtol=1e-12)
        np.testing.assert_allclose(z, ez, rtol=1e-9, atol=1e-12, equal_nan=True)
        np.testing.assert_allclose(rr, er, rtol=1e-12, equal_nan=True)
        assert np.array_equal(n, en)
        np.testing.assert_allclose([mlr, mz], [emlr, emz], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("mode", ["strong", "replicas", "strong8"])
def test_bench_two_ranks_smoke(mode):
    """bench.py's N > 1 code path end to end (launcher contract, sharding of the A pass and of the
    gonosomal passes, collectives, replica predict, JSON line) with two gloo ranks on one device and
    a small problem; and the --replicas throughput mode (one whole reference per rank)."""
    import json
    import subprocess
    env = dict(os.environ, WCX_DIST_BACKEND="gloo", WCX_BENCH_SPINUP_STEPS="1", MASTER_ADDR="127.0.0.1")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    n = "8" if mode == "strong8" else "2"          # (8 ranks: the driver's largest launch, uneven shards)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", n,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", n, "--steps", "2", "--warmup", "1", "--binsize", "100000", "--samples", "40",
           "--refsize", "100"] + (["--replicas"] if mode == "replicas" else [])
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                       # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == int(n) and d["steps"] == 2 and d["value"] > 0 and "cpu_baseline" not in d
    assert d["roofline"]["frac"] > 0
    assert d["scaling"] == ("weak" if mode == "replicas" else "strong")
    assert d["verified"]["mismatches"] == 0 and d["verified"]["rows"] > 100


def _worker_sym(rank, world, port, X, cum, k, ids, q, ovf_rank=-1):
    try:
        _worker_sym_body(rank, world, port, X, cum, k, ids, q, ovf_rank)
    except Exception as e:            # (the parent must not wait for its timeout)
        import traceback
        q.put((rank, "error", "{}\n{}".format(e, traceback.format_exc())))


def _worker_sym_body(rank, world, port, X, cum, k, ids, q, ovf_rank=-1):
    sys.path.insert(0, ROOT)
    os.environ["WCX_SYM_SHARD_MIN"] = "2"
    if rank == ovf_rank:
        os.environ["WCX_SYM_TEST_POOL_OVF"] = "1"      # this rank reports a record-pool overflow
    import torch
    import torch.distributed as dist
    from wisecondorx_amd import _lib
    from wisecondorx_amd import dist as wd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    be = wd.GpuBackend(ctx)
    B = X.shape[0]
    b, e = wd.row_shard(rank, world, B)
    pad = wd.max_shard_rows(world, B)
    local = torch.zeros((pad, X.shape[1]), dtype=torch.float64, device=dev)
    local[:e - b] = torch.from_numpy(np.ascontiguousarray(X[b:e])).to(dev)
    wd.newref_sym_sharded.last_records = None
    idx, dd, nr, Xs = wd.newref_sym_sharded(local, B, cum, k, ids, be, rank, world)
    ctx.sync()
    st = ctx.topk_stats()
    q.put((rank, idx.cpu().numpy().copy(), dd.cpu().numpy().copy(), nr.cpu().numpy().copy(),
           wd.newref_sym_sharded.last_records, st["fallback_rows"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_row_sharded_symmetric_sweep(world):
    """dist.newref_sym_sharded with the real kernels, all ranks on one device over gloo: the tile pairs of
    the symmetric sweep dealt out to the ranks (each pair once, both directions), the hit records routed
    to the rows' owners by ONE all-to-all -- every rank's row block bit-identical to the oracle's (and so
    to the one-process build); the symmetric path must really have run on every rank."""
    import torch.multiprocessing as mp
    from oracle import c_oracle as CO
    from oracle import wcx_oracle as O
    from wisecondorx_amd.synth import corrected_matrix
    X, mbpc, cum = corrected_matrix([7100, 6600, 6100, 5400, 4700, 3901], 256, seed=29)    # 33 801 rows
    X = np.asfortranarray(X)
    B, k, ids = cum[-1], 64, [3, 1, 7, 0, 22, 39, 255, 100]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker_sym, args=(r, world, port, X, cum, k, ids, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in res:
        assert r[4] is not None and r[4][0] > 0 and r[4][1] > 0, "rank {}: the symmetric sweep did not run".format(r[0])
    assert sum(r[4][0] for r in res) == sum(r[4][1] for r in res)          # every record arrived somewhere
    ei, ed = CO.get_reference_rows_threaded(np.ascontiguousarray(X.T), cum, 0, B, k)
    gi, gd = np.concatenate([r[1] for r in res]), np.concatenate([r[2] for r in res])
    bad = np.flatnonzero((gi != ei).any(axis=1) | (gd != ed).any(axis=1))
    assert bad.size == 0, "{} of {} rows differ (first {})".format(bad.size, B, bad[:5])
    gnr = np.concatenate([r[3] for r in res])
    for lo in (0, B // 2 - 500, B - 1000):            # (the NumPy oracle walks the rows one by one)
        with np.errstate(all="ignore"):
            enr = O.null_ratios(X, ei[lo:lo + 1000], lo, lo + 1000, ids)
        np.testing.assert_allclose(gnr[lo:lo + 1000], enr, rtol=1e-12, atol=1e-13)
    assert max(r[5] for r in res) <= 64              # rows the exact kernel had to redo, per rank


def test_row_sharded_symmetric_sweep_void_exchange():
    """A record-pool overflow on ONE rank (data-dependent in production, forced here) must not leave its
    peers waiting in the all-to-all nor cost a hit: the counts of -1 it sends make the exchange void on
    every rank alike, and each rank redoes its own rows with the exact kernel -- same bits as ever."""
    import torch.multiprocessing as mp
    from oracle import c_oracle as CO
    from wisecondorx_amd.synth import corrected_matrix
    X, mbpc, cum = corrected_matrix([7100, 6600, 6100, 5400, 4700, 3901], 256, seed=29)
    X = np.asfortranarray(X)
    B, k, ids, world = cum[-1], 64, [3, 1, 7, 0, 22, 39, 255, 100], 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker_sym, args=(r, world, port, X, cum, k, ids, q, 1)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in procs], key=lambda t: t[0])
    assert all(not (isinstance(r[1], str) and r[1] == "error") for r in res), [r[2] for r in res if r[1] == "error"]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in res:
        assert r[4] is not None and r[4][1] == -1, "rank {}: the exchange was not void".format(r[0])
    ei, ed = CO.get_reference_rows_threaded(np.ascontiguousarray(X.T), cum, 0, B, k)
    gi, gd = np.concatenate([r[1] for r in res]), np.concatenate([r[2] for r in res])
    bad = np.flatnonzero((gi != ei).any(axis=1) | (gd != ed).any(axis=1))
    assert bad.size == 0, "{} of {} rows differ (first {})".format(bad.size, B, bad[:5])
    assert min(r[5] for r in res) > B // 4            # every rank redid its rows exactly
