#!/usr/bin/env python3
"""`WisecondorX convert / newref / gender / predict` -- the reference's command line
(main.py:302-498: same sub-commands, flags, defaults and validation rules) driving the MI355X
hot path.  Additions: `--gpus` (newref: row parts are spread over that many devices; the
reference's --cpus is accepted and ignored).  `--plot` needs R/plotter.R and is outside the
hot path: it logs a warning and is skipped."""
import argparse
import logging
import os
import random
import sys
import warnings

import numpy as np

from . import npz_io, prep
from .overall_tools import gender_correct, predict_gender, scale_sample


# --------------------------------------------------------------------------- convert
def tool_convert(args):
    logging.info("Starting conversion")
    try:
        import pysam  # noqa: F401
    except ImportError:
        logging.critical("convert needs pysam/htslib (BAM/CRAM parsing); it is I/O-bound host "
                         "work outside the MI355X hot path and pysam is not installed here")
        sys.exit()
    from .convert_tools import convert_reads
    sample, qual_info = convert_reads(args)
    np.savez_compressed(args.outfile, binsize=args.binsize, sample=sample, quality=qual_info)
    logging.info("Finished conversion")


# --------------------------------------------------------------------------- newref
def _read_totals(sample):
    """(all reads, reads on chrY (key "24")) of one sample."""
    return float(sum(v.sum() for v in sample.values())), float(sample["24"].sum())   # (the method: a third of
                                                                                       #  np.sum's call overhead)


def _y_fractions(samples, totals=None):
    """Share of the reads of each sample that fall on chrY.  totals: _read_totals of every sample if the
    loader threads already took them (12 500 small reductions: 70 ms in one thread)."""
    totals = np.array(totals if totals is not None else [_read_totals(s) for s in samples], dtype=np.float64)
    return totals[:, 1] / totals[:, 0]


def _plot_yfrac(path, y_fractions, grid, density):
    import matplotlib
    matplotlib.use("Agg")
    from matplotlib import pyplot
    figure = pyplot.figure(figsize=(16, 6))
    axes = figure.add_subplot(111)
    axes.hist(y_fractions, bins=100, density=True)
    axes.plot(grid, density, "r-", label="Gaussian mixture fit")
    axes.set_xlim(grid[0], grid[-1])
    axes.legend(loc="best")
    figure.savefig(path)


def train_gender_model(args, samples, totals=None):
    """Gender of every reference sample from its Y-read fraction (newref_tools.py:21-68): with
    --yfrac the cut-off is given; otherwise a two-component Gaussian mixture is fitted and the
    cut-off is the first local minimum of its density on [0, 0.02] (same mixture settings and
    grid as the reference, so the same cut-off)."""
    y_fractions = _y_fractions(samples, totals)
    cut_off = args.yfrac
    if cut_off is None:
        from scipy.signal import argrelextrema
        from sklearn.mixture import GaussianMixture
        mixture = GaussianMixture(n_components=2, covariance_type="full", reg_covar=1e-99,
                                  max_iter=10000, tol=1e-99).fit(y_fractions[:, None])
        grid = np.linspace(0, 0.02, 5000)
        density = np.exp(mixture.score_samples(grid[:, None]))
        if getattr(args, "plotyfrac", None) is not None:
            _plot_yfrac(args.plotyfrac, y_fractions, grid, density)
            logging.info("Image written to {}, now quitting ...".format(args.plotyfrac))
            sys.exit()
        minima = argrelextrema(density, np.less)[0]
        if not len(minima):
            logging.critical("No local minimum in the Y-fraction mixture density: the genders "
                             "cannot be separated automatically, pass --yfrac")
            sys.exit()
        cut_off = grid[minima[0]]
        logging.info("Determined --yfrac cutoff: {}".format(str(round(cut_off, 4))))
    genders = np.full(len(samples), None, dtype=object)
    genders[y_fractions > cut_off] = "M"
    genders[y_fractions < cut_off] = "F"
    return genders.tolist(), cut_off


class _DevMatrix:
    """A device pointer as a zero-copy torch tensor (`torch.as_tensor` reads __cuda_array_interface__)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(int(v) for v in shape), "typestr": "<f8",
                                         "data": (int(ptr), False), "version": 2}


def _build_sub_reference_sharded(args, gender, total_mask, bins_per_chr, dc, sel, frozen, rd):
    """One pass of a multi-process build (newref --gpus N, one process per GPU; rd = rank, world,
    dist backend object): every rank prepares the pass on its own device, keeps ITS rows of the corrected
    matrix (newref_tools._get_part over the rows, newref_tools.py:244-247), and the ranks exchange them
    with ONE all-gather over RCCL / xGMI (dist.newref_sharded); each searches its own target rows and the
    finished row blocks are all-gathered so that rank 0 can write the file (newref_control.py:90-109
    does the same split into parts, through files)."""
    import ctypes as C
    import torch
    from . import _lib, dist as wd
    rank, world, backend = rd
    ctx = backend.ctx
    p = prep.prepare_dev(dc, sel, gender, total_mask, bins_per_chr, frozen=frozen)
    p.pop("X")
    S = int(p.pop("n_samples"))
    cum = [int(v) for v in p["masked_bins_per_chr_cum"]]
    B, k = cum[-1], int(args.refsize)
    sample_ids = np.asarray(random.sample(range(S), min(S, 100)), dtype=np.int32)   # newref_tools.py:214-217
    dX = C.c_void_p()
    _lib.check(ctx.lib.wcx_pca_corrected_dev(ctx.h, C.byref(dX)))
    Xs = torch.as_tensor(_DevMatrix(dX.value, (S, B)), device=torch.device("cuda", ctx.device))
    rb, re_ = wd.row_shard(rank, world, B)
    local = torch.zeros((wd.max_shard_rows(world, B), S), dtype=torch.float64, device=Xs.device)
    local[:re_ - rb] = Xs[:, rb:re_].t()
    if gender == "A":
        # (the symmetric sweep's tile pairs dealt out to the ranks + one all-to-all of the hit records;
        #  where the library has no symmetric sweep for the shape it searches the row range itself)
        idx_l, dist_l, nr_l, _ = wd.newref_sym_sharded(local, B, cum, k, sample_ids, backend, rank, world)
        idx, dist_, nr = wd.gather_reference3(idx_l, dist_l, nr_l, B, world, backend)
    else:
        dev = Xs.device
        full = (torch.empty((B, k), dtype=torch.int32, device=dev),
                torch.empty((B, k), dtype=torch.float64, device=dev),
                torch.empty((B, len(sample_ids)), dtype=torch.float64, device=dev))
        idx, dist_, nr = wd.newref_gonosomal_sharded(local, B, cum, k, sample_ids, backend, rank, world, full)
    out = dict(p)
    out["binsize"] = args.binsize
    if rank == 0:
        ct = cum[21] if len(cum) > 22 else 0
        if gender != "A" and 0 < ct < B:
            # (as get_reference_dev: the autosomal rows of a gonosomal pass are the constants 0 / 1 and stay
            #  on the device; the writer stores them as a few deflate blocks -- npz_io.PrefixConst)
            out["indexes"] = npz_io.PrefixConst(ct, 0, idx[ct:].cpu().numpy())
            out["distances"] = npz_io.PrefixConst(ct, 1.0, dist_[ct:].cpu().numpy())
            out["null_ratios"] = nr.cpu().numpy()
        else:
            out["indexes"], out["distances"], out["null_ratios"] = (t.cpu().numpy() for t in (idx, dist_, nr))
    else:
        torch.cuda.synchronize()
    return out


def build_sub_reference(args, samples, gender, total_mask, bins_per_chr, contexts, dc=None, sel=None,
                        rd=None):
    """One of the A / F / M passes: tool_newref_prep + tool_newref_main + tool_newref_post
    (newref_control.py:24-189) without the temp-file round trips.
    dc / sel: the cohort's device-resident counts (prep.DeviceCounts) and this pass's sample
    indexes -- normalisation, masking and the PCA then run from HBM, and on one GPU the corrected
    matrix never leaves it."""
    from . import newref_tools
    # PCA on the device.  DEFAULT = upstream (newref_control.py:48-54): the PCA-distance filter of a
    # gonosomal pass may drop autosomal bins the finished A reference still holds (the shared mask
    # is mutated in place after the A pass saved its copy).  Upstream's predict then raises
    # IndexError on such a reference (predict_control.py:50 on results_nr; pinned by
    # tests/golden/mask_skew.npz) and so does ours (tool_test).  --aligned-masks freezes the
    # autosomal part of the mask in the gonosomal passes instead (see prep.prepare).
    frozen = 0
    if gender != "A" and getattr(args, "aligned_masks", False):
        frozen = int(np.sum(bins_per_chr[:22]))
    if rd is not None:
        return _build_sub_reference_sharded(args, gender, total_mask, bins_per_chr, dc, sel, frozen, rd)
    n_parts = len(contexts)
    if dc is not None:
        p = prep.prepare_dev(dc, sel, gender, total_mask, bins_per_chr, frozen=frozen,
                             want_host_X=n_parts > 1)
    else:
        p = prep.prepare(samples, gender, total_mask, bins_per_chr, ctx=contexts[0], frozen=frozen)
    X = p.pop("X")
    n_samples = p.pop("n_samples") if "n_samples" in p else X.shape[1]
    cum = [int(v) for v in p["masked_bins_per_chr_cum"]]
    sample_ids = random.sample(range(n_samples), min(n_samples, 100))     # newref_tools.py:214-217
    if X is None:
        parts = [newref_tools.get_reference_dev(contexts[0], n_samples, cum, args.refsize, sample_ids)]
    else:
        parts = newref_tools.get_reference_parts(X, cum, args.refsize, n_parts, sample_ids, contexts)
    out = dict(p)
    out["binsize"] = args.binsize
    cat = (lambda seq: seq[0]) if len(parts) == 1 else np.concatenate     # (no 0.8 GB copy for one part)
    out["indexes"] = cat([q[0] for q in parts])
    out["distances"] = cat([q[1] for q in parts])
    out["null_ratios"] = cat([q[2] for q in parts])
    return out


def _newref_rank(rank, world, args, port, rnd_state):
    """One process per GPU of `newref --gpus N`: torch.distributed over RCCL ("nccl"; WCX_DIST_BACKEND=gloo
    with WCX_DIST_SHARE_DEVICE=1 lets the tests run several ranks on one device)."""
    import torch
    import torch.distributed as dist
    from . import _lib, dist as wd
    backend_name = os.environ.get("WCX_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    dev_index = rank % n_dev if os.environ.get("WCX_DIST_SHARE_DEVICE") else rank
    torch.cuda.set_device(dev_index)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend_name == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
    else:
        dist.init_process_group(backend_name, rank=rank, world_size=world)
    random.setstate(rnd_state)          # every rank draws the parent's sequence (the null samples)
    logging.basicConfig(format="[%(levelname)s - %(asctime)s]: %(message)s", datefmt="%Y-%m-%d %H:%M:%S",
                        level=getattr(logging, str(args.loglevel).upper(), None) if rank == 0 else logging.ERROR)
    try:
        ctx = _lib.Context(dev_index, torch.cuda.current_stream().cuda_stream)
        _newref_body(args, [ctx], (rank, world, wd.GpuBackend(ctx)))
    finally:
        dist.destroy_process_group()


def _keep_freed_memory_in_the_heap():
    """glibc serves allocations above 128 KB by mmap and returns them by munmap: every imported sample file
    costs a handful of those (0.8 MB inflate buffers), and unmapping -- address-space lock, TLB shootdowns on
    every core that runs one of the loader threads -- is what the threads queue for (500 files at 15 kb:
    0.33 -> 0.24 s).  Raise the two thresholds for this process: M_MMAP_THRESHOLD (-3) 8 MB, M_TRIM_THRESHOLD
    (-1) 256 MB; the result tables (hundreds of MB) stay mmap'd -- served from the heap they made the passes
    three times slower, measured -- and go back to the system when freed."""
    try:
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 8 << 20)
        libc.mallopt(-1, 256 << 20)
    except (OSError, AttributeError):
        pass


def tool_newref(args):
    logging.info("Creating new reference")
    _keep_freed_memory_in_the_heap()
    if args.yfrac is not None and (args.yfrac < 0 or args.yfrac > 1):
        logging.critical("Parameter --yfrac should be a positive number lower than or equal to 1")
        sys.exit()
    from . import _lib
    n_gpus = max(1, int(getattr(args, "gpus", 1) or 1))
    if n_gpus > 1 and not os.environ.get("WCX_NEWREF_ONE_PROCESS"):
        # one process per GPU; the row shards of the corrected matrix meet in ONE all-gather per pass
        import socket
        import torch.multiprocessing as mp
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        mp.spawn(_newref_rank, args=(n_gpus, args, port, random.getstate()), nprocs=n_gpus, join=True)
        return
    # (the device contexts come up -- HIP runtime, streams, workspaces: ~80 ms -- beside the sample import)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=1) as ex:
        _newref_body(args, ex.submit(lambda: [_lib.default_context(d) for d in range(n_gpus)]), None)


def _newref_body(args, contexts, rd):
    """tool_newref (main.py:43-138).  rd = None: one process (all devices of `contexts`); rd = (rank,
    world, backend): this process is one rank of a multi-process build."""
    rank, world = (rd[0], rd[1]) if rd else (0, 1)

    samples = []
    logging.info("Importing data ...")
    totals = []

    def load_one(infile):                       # (unzip + unpickle release the GIL for most of it)
        sample, binsize = npz_io.load_sample(infile)
        sample = scale_sample(sample, binsize, args.binsize)
        return sample, int(binsize), _read_totals(sample)      # (read totals for the gender model, while hot)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=npz_io._LOAD_THREADS) as ex:
        for infile, (sample, binsize, tot) in zip(args.infiles, ex.map(load_one, args.infiles)):
            logging.info("Loading: {}".format(infile))
            logging.info("Binsize: {}".format(binsize))
            samples.append(sample)
            totals.append(tot)
    samples = np.array(samples)
    if world > 1:
        # the gender model fits a Gaussian mixture from a random start: rank 0 decides for everybody
        import torch.distributed as dist
        box = [train_gender_model(args, samples, totals) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        genders, trained_cutoff = box[0]
    else:
        genders, trained_cutoff = train_gender_model(args, samples, totals)

    if genders.count("F") < 5 and args.nipt:
        logging.warning("A NIPT reference should have at least 5 female feti samples. "
                        "Removing --nipt flag.")
        args.nipt = False
    if not args.nipt:
        for i, sample in enumerate(samples):
            samples[i] = gender_correct(sample, genders[i])

    g = np.array(genders)
    if hasattr(contexts, "result"):
        contexts = contexts.result()
    # the (gender-corrected) counts go to the device once; masks, every pass's normalisation and
    # PCA read them there.  Non-integer counts (not something `convert` writes) take the host path.
    dc = None
    try:
        dc = prep.DeviceCounts(contexts[0], samples)
    except (TypeError, ValueError, RuntimeError) as e:      # non-integer counts, a count beyond int32,
        logging.info("Host-side masks / normalisation: {}".format(e))   # no room in HBM: the host path works
    if rd is not None and dc is None:
        logging.critical("newref --gpus N needs the device-resident counts (integer counts that fit the device)")
        sys.exit(1)
    sel_of = {"A": np.arange(len(genders)), "F": np.flatnonzero(g == "F"), "M": np.flatnonzero(g == "M")}
    get_mask = (lambda k: dc.get_mask(sel_of[k])) if dc is not None else \
        (lambda k: prep.get_mask(samples[sel_of[k]]))
    total_mask, bins_per_chr = get_mask("A")
    if genders.count("F") > 4:
        total_mask = total_mask & get_mask("F")[0]
    if genders.count("M") > 4 and not args.nipt:
        total_mask = total_mask & get_mask("M")[0]

    final_ref = {"has_female": False, "has_male": False}
    # the reference file is written WHILE it is built: a pass's tables go to the writer's threads as
    # soon as they are on the host and land in the file beside the next pass's device work
    writer = npz_io.NpzWriter(args.outfile) if rank == 0 else None
    written = set()

    # The three tables of a pass (0.8 GB at 15 kb) are not kept until the end: once a pass's members are
    # with the writer and its QC metrics taken (ref_qc.compute_metrics, on a worker thread), this dict
    # lets go of them -- the writer frees them on its threads when they are on disk, beside the next
    # pass's device work, instead of 2.5 GB being unmapped when the build returns.
    will_f = genders.count("F") > 4
    will_m = (not args.nipt) and genders.count("M") > 4
    qc_metrics, qc_jobs = {}, []
    from concurrent.futures import ThreadPoolExecutor as _Pool
    qc_pool = _Pool(max_workers=1)

    def stream_out(suf=None):
        if writer is not None:
            for k_, v_ in list(final_ref.items()):
                if k_ not in written and k_ not in ("has_female", "has_male"):
                    writer.add(k_, v_)
                    written.add(k_)
            writer.flush_async()
            if suf is not None:
                def take_metrics_and_release():
                    from .ref_qc import compute_metrics
                    if suf != "" or not (will_f or will_m):      # (ref_qc.py:11-20: the autosomal set only
                        qc_metrics[suf] = compute_metrics(final_ref, suf)   #  gets a verdict on its own)
                    for name in ("indexes", "distances", "null_ratios"):
                        final_ref.pop(name + suf, None)
                qc_jobs.append(qc_pool.submit(take_metrics_and_release))
    try:
        if len(genders) > 9:
            logging.info("Starting autosomal reference creation ...")
            sub = build_sub_reference(args, samples, "A", total_mask, bins_per_chr, contexts, dc, sel_of["A"], rd)
            final_ref.update({k: v for k, v in sub.items() if k != "gender"})
            del sub
            stream_out("")
        else:
            logging.critical("Provide at least 10 samples to enable the generation of a reference.")
            sys.exit()
        if genders.count("F") > 4:
            logging.info("Starting female gonosomal reference creation ...")
            sub = build_sub_reference(args, samples[g == "F"], "F", total_mask, bins_per_chr, contexts, dc,
                                      sel_of["F"], rd)
            final_ref["has_female"] = True
            final_ref.update({k + ".F": v for k, v in sub.items() if k != "gender"})
            del sub
            stream_out(".F")
        else:
            logging.warning("Provide at least 5 female samples to enable normalization of female gonosomes.")
        if not args.nipt:
            if genders.count("M") > 4:
                logging.info("Starting male gonosomal reference creation ...")
                sub = build_sub_reference(args, samples[g == "M"], "M", total_mask, bins_per_chr, contexts, dc,
                                          sel_of["M"], rd)
                final_ref["has_male"] = True
                final_ref.update({k + ".M": v for k, v in sub.items() if k != "gender"})
                del sub
                stream_out(".M")
            else:
                logging.warning("Provide at least 5 male samples to enable normalization of male gonosomes.")
    except BaseException:
        if writer is not None:
            writer.abort()
        raise
    finally:
        # (also on an error in any pass: the device counts, the PCA stage and the buffers must not
        # outlive the call while the context does)
        if dc is not None:
            dc.close()
            contexts[0].lib.wcx_pca_end(contexts[0].h)
            contexts[0].release_buffers()
    # Mask skew is judged by EVERY rank (each prepared every pass itself, so each holds the three masks): a
    # refused reference ends all ranks with the same exit status, not rank 0 alone.
    n_aut = int(np.sum(final_ref["bins_per_chr"]))
    skewed = []
    for ap in (".F", ".M"):
        if "mask" + ap in final_ref and not np.array_equal(final_ref["mask" + ap][:n_aut],
                                                           final_ref["mask"]):
            # the reference has the same latent skew as upstream's (newref_control.py:51-54 mutates
            # the shared mask after the A pass kept its copy) and cannot be aligned at predict time
            skewed.append("the PCA-distance filter of the {} pass dropped {} autosomal bin(s) the "
                          "autosomal reference still holds".format(
                              ap[1:], int(np.sum(final_ref["mask"]) -
                                          np.sum(final_ref["mask" + ap][:n_aut]))))
    refuse = bool(skewed) and not getattr(args, "reference_mask_skew", False)
    if rank != 0:
        qc_pool.shutdown(wait=False)
        if refuse:
            sys.exit(1)
        return                      # rank 0 holds the gathered tables and writes the file
    closing = None
    try:
        final_ref["is_nipt"] = args.nipt
        final_ref["trained_cutoff"] = trained_cutoff
        if skewed and not refuse:
            logging.warning("{} (upstream behaviour, kept on request): predict cannot use this "
                            "reference -- rebuild with --aligned-masks".format("; ".join(skewed)))
        elif skewed:
            # upstream writes such a reference and its predict then dies with an IndexError
            # (predict_control.py:50, tests/golden/mask_skew.npz); an unusable file is not written here
            # (a deliberate deviation from upstream's CLI, which exits 0 with the unusable file)
            logging.critical("{}: no predict can use such a reference (upstream's raises IndexError at "
                             "predict_control.py:50), so it is NOT written.  Rebuild with --aligned-masks "
                             "(keeps the autosomal masks of the three passes equal), or with "
                             "--reference-mask-skew to write upstream's file as it is".format(
                                 "; ".join(skewed)))
            writer.abort()
            qc_pool.shutdown(wait=False)
            sys.exit(1)
        for k_ in ("has_female", "has_male", "is_nipt", "trained_cutoff"):
            writer.add(k_, final_ref[k_])
        # the last writes and the sync run beside the QC of the tables (main.py:134-135)
        from concurrent.futures import ThreadPoolExecutor
        from .ref_qc import qc_reference
        with ThreadPoolExecutor(max_workers=1) as ex:
            closing = ex.submit(writer.close)
            logging.info("Running QC on the newly created reference...")
            try:
                for j_ in qc_jobs:
                    j_.result()
                qc_pool.shutdown(wait=True)
                qc_reference(final_ref, qc_metrics)
            finally:
                closing.result()
    except BaseException:
        # anything between the passes and the close (the skew check, a missing key, a failing QC
        # job): no open descriptor, no writer threads, no multi-GB temporary file left behind
        if closing is None:
            writer.abort()
        qc_pool.shutdown(wait=False)
        raise
    logging.info("Finished creating reference")


# --------------------------------------------------------------------------- gender / predict
def output_gender(args):
    ref_file = npz_io.load_reference(args.reference, defer=("",))       # (only the small members are read)
    sample, _ = npz_io.load_sample(args.infile)
    print("male" if predict_gender(sample, ref_file["trained_cutoff"]) == "M" else "female")


def _check_mask_alignment(ref_file, ap):
    """The merged result of predict is inflated with mask{ap}: its autosomal part must select the bins
    the autosomal reference holds (see build_sub_reference)."""
    n_aut_masked = int(np.sum(ref_file["mask"]))
    n_aut = int(np.sum(ref_file["bins_per_chr"]))
    mask_aut, mask_gon_aut = np.asarray(ref_file["mask"])[:n_aut], np.asarray(ref_file["mask" + ap])[:n_aut]
    if int(np.sum(mask_gon_aut)) != n_aut_masked:
        # Upstream fails on the same input: the merged null-ratio table has sum(mask{ap}) rows but
        # ref_sizes one entry per merged result, so get_post_processed_result raises "IndexError:
        # boolean index did not match" (predict_control.py:50; run of the reference itself recorded in
        # tests/golden/mask_skew.npz).  Same outcome here, with the reason spelled out.
        logging.critical("Reference mask{} holds {} autosomal bins but the autosomal reference {}: "
                         "the reference was built with a PCA-distance filter skew "
                         "(newref_control.py:48-54) and cannot be aligned (upstream raises IndexError "
                         "at predict_control.py:50 for it); rebuild it with newref --aligned-masks".format(
                             ap, int(np.sum(mask_gon_aut)), n_aut_masked))
        sys.exit(1)
    if not np.array_equal(mask_gon_aut, mask_aut):
        logging.warning("Reference mask{} keeps the same NUMBER of autosomal bins as the autosomal "
                        "reference but at {} different positions: autosomal results are reported "
                        "at the positions of mask{} (as upstream does)".format(
                            ap, int(np.sum(mask_gon_aut != mask_aut)), ap))


def tool_test(args):
    logging.info("Starting CNA prediction")
    if not args.bed and not args.plot:
        logging.critical("No output format selected. Select at least one of the supported output "
                         "formats (--bed, --plot)")
        sys.exit()
    if args.zscore <= 0:
        logging.critical("Parameter --zscore should be a strictly positive number")
        sys.exit()
    if args.beta is not None and (args.beta <= 0 or args.beta > 1):
        logging.critical("Parameter --beta should be a strictly positive number lower than or equal to 1")
        sys.exit()
    if args.alpha <= 0 or args.alpha > 1:
        logging.critical("Parameter --alpha should be a strictly positive number lower than or equal to 1")
        sys.exit()
    if args.plot and not args.bed:
        logging.critical("--plot needs R (include/plotter.R of the reference), which is outside the "
                         "MI355X hot path: no output would be written. Add --bed.")
        sys.exit()
    from . import predict_tools as pt
    from .predict_output import generate_output_tables

    if getattr(args, "batch", None):
        pairs = [(args.infile, args.outid)]
        with open(args.batch) as fh:
            for line in fh:
                f = line.split()
                if len(f) >= 2 and not line.startswith("#"):
                    pairs.append((f[0], f[1]))
        return tool_test_batch(args, pairs)

    logging.info("Importing data ...")
    # (the gonosomal tables -- a third of the file each -- are read once the sample's gender says which)
    ref_file = npz_io.load_reference(args.reference, defer=(".F", ".M"))
    sample, sample_binsize = npz_io.load_sample(args.infile)
    n_reads = sum([sum(sample[x]) for x in sample.keys()])
    sample = scale_sample(sample, int(sample_binsize), int(ref_file["binsize"]))

    gender = predict_gender(sample, ref_file["trained_cutoff"])
    if not ref_file["is_nipt"]:
        if args.gender:
            gender = args.gender
        sample = gender_correct(sample, gender)
        ref_gender = gender
    else:
        if args.gender:
            gender = args.gender
        ref_gender = "F"

    # the gonosomal set this sample will use (the rules below, decided early): its tables are read on a
    # worker thread beside the autosomal normalisation
    early = ref_gender
    if not ref_file["is_nipt"]:
        if not ref_file["has_male"] and gender == "M":
            early = "F"
        elif not ref_file["has_female"] and gender == "F":
            early = "M"
    from concurrent.futures import ThreadPoolExecutor
    gon_loader = ThreadPoolExecutor(max_workers=1)
    gon_loaded = gon_loader.submit(npz_io.ensure_loaded, ref_file, ".{}".format(early))
    gon_loader.shutdown(wait=False)

    cache = {}
    logging.info("Normalizing autosomes ...")
    res_a = pt.normalize(args, sample, ref_file, "A", cache)
    if not ref_file["is_nipt"]:
        if not ref_file["has_male"] and gender == "M":
            logging.warning("This sample is male, whilst the reference is created with fewer than 5 "
                            "males. The female gonosomal reference will be used for X predictions.")
            ref_gender = "F"
        elif not ref_file["has_female"] and gender == "F":
            logging.warning("This sample is female, whilst the reference is created with fewer than 5 "
                            "females. The male gonosomal reference will be used for XY predictions.")
            ref_gender = "M"
    logging.info("Normalizing gonosomes ...")
    ap = ".{}".format(ref_gender)
    gon_loaded.result()
    nr_aut = ref_file["null_ratios"]
    nr_gon = ref_file["null_ratios" + ap][len(nr_aut):]
    res_g = pt.normalize(args, sample, ref_file, ref_gender, cache)

    rem_input = {
        "args": args, "binsize": int(ref_file["binsize"]), "n_reads": n_reads,
        "ref_gender": ref_gender, "gender": gender, "mask": ref_file["mask" + ap],
        "bins_per_chr": ref_file["bins_per_chr" + ap],
        "masked_bins_per_chr": ref_file["masked_bins_per_chr" + ap],
        "masked_bins_per_chr_cum": ref_file["masked_bins_per_chr_cum" + ap],
    }
    m_lr = res_a[4]
    _check_mask_alignment(ref_file, ap)
    r, z, w, ref_sizes, weights_ok = pt.merge_autosomes_gonosomes(res_a, res_g)
    if not weights_ok:       # main.py:252-256
        logging.warning("Non-numeric values found in weights -- reference too small. "
                        "Circular binary segmentation and z-scoring will be unweighted")
    m = max(nr_aut.shape[1], nr_gon.shape[1])
    nr = np.full((len(nr_aut) + len(nr_gon), m), np.nan)      # ragged rows padded with NaN
    nr[:len(nr_aut), :nr_aut.shape[1]] = nr_aut
    nr[len(nr_aut):, :nr_gon.shape[1]] = nr_gon
    results = {"results_r": r, "results_z": z, "results_w": w}
    for key in results:
        results[key] = pt.get_post_processed_result(args, results[key], ref_sizes, rem_input)
    # null ratios: inflated once and kept on the device (rows of bins with ratio 0 are skipped by
    # the segment-z kernel, so the per-sample zeroing of predict_control.py:50-52 is a no-op)
    off = np.concatenate(([0], np.cumsum(rem_input["bins_per_chr"]))).astype(int)
    nr_full = pt.inflate_results(nr, rem_input)
    pt.attach_null_matrix([nr_full[off[c]:off[c + 1]] for c in range(len(off) - 1)])
    results["results_nr"] = pt.ATTACHED
    pt.log_trans(results, m_lr)
    if args.blacklist:
        logging.info("Applying blacklist ...")
        pt.apply_blacklist(rem_input, results)
    logging.info("Executing circular binary segmentation ...")
    results["results_c"] = pt.exec_cbs(rem_input, results)
    if args.bed:
        logging.info("Writing tables ...")
        generate_output_tables(rem_input, results)
    if args.plot:
        logging.warning("--plot needs R (include/plotter.R of the reference); plotting is outside "
                        "the MI355X hot path and is skipped")
    logging.info("Finished prediction")
    return results


# --------------------------------------------------------------------------- predict, batches
def _batch_worker(rank, world, args, pairs):
    """One process per GPU: this rank's stripe of the batch (dist.stripe -- no collective on this
    path, docs/include/pipeline/predict.sh:21 loops the samples), device-resident end to end
    (dist.predict_batch_dev): counts -> coverage normalisation + PCA projection -> three
    normalisation passes for autosomes and gonosomes -> merge / post-processing -> ONE batched CBS and
    segment-z call per chunk -> the reference's tables per sample."""
    import torch
    from . import _lib, dist as wd, predict_tools as pt
    from .predict_output import generate_output_tables
    n_dev = torch.cuda.device_count()
    dev_index = rank % n_dev if os.environ.get("WCX_DIST_SHARE_DEVICE") else rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    ctx = _lib.Context(dev_index, torch.cuda.current_stream().cuda_stream)
    be = wd.GpuBackend(ctx)
    mine = wd.stripe(pairs, rank, world)
    if not mine:
        return
    ref_file = npz_io.load_reference(args.reference, defer=(".F", ".M"))
    binsize = int(ref_file["binsize"])
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
    A = {"idx": tt(ref_file["indexes"]), "dist": tt(ref_file["distances"]), "nr": tt(ref_file["null_ratios"]),
         "cum": [int(v) for v in ref_file["masked_bins_per_chr_cum"]]}
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=npz_io._THREADS) as ex:
        loaded = list(ex.map(lambda p_: npz_io.load_sample(p_[0]), mine))
    groups = {}
    for (infile, outid), (sample, sample_binsize) in zip(mine, loaded):
        n_reads = sum([sum(sample[x]) for x in sample.keys()])
        sample = scale_sample(sample, int(sample_binsize), binsize)
        gender = predict_gender(sample, ref_file["trained_cutoff"])
        if not ref_file["is_nipt"]:
            gender = args.gender or gender
            sample = gender_correct(sample, gender)
            ref_gender = gender
            if not ref_file["has_male"] and gender == "M":
                ref_gender = "F"
            elif not ref_file["has_female"] and gender == "F":
                ref_gender = "M"
        else:
            gender = args.gender or gender
            ref_gender = "F"
        groups.setdefault(ref_gender, []).append((infile, outid, sample, gender, n_reads))
    chunk = max(1, int(getattr(args, "batch_size", 96) or 96))
    for ref_gender, items in groups.items():
        ap = ".{}".format(ref_gender)
        npz_io.ensure_loaded(ref_file, ap)
        _check_mask_alignment(ref_file, ap)
        G = {"idx": tt(ref_file["indexes" + ap]), "dist": tt(ref_file["distances" + ap]),
             "nr": tt(ref_file["null_ratios" + ap]),
             "cum": [int(v) for v in ref_file["masked_bins_per_chr_cum" + ap]]}
        rem = {"args": args, "binsize": binsize, "ref_gender": ref_gender, "mask": ref_file["mask" + ap],
               "bins_per_chr": ref_file["bins_per_chr" + ap],
               "masked_bins_per_chr": ref_file["masked_bins_per_chr" + ap],
               "masked_bins_per_chr_cum": ref_file["masked_bins_per_chr_cum" + ap]}
        off = np.concatenate(([0], np.cumsum(rem["bins_per_chr"]))).astype(int)
        prep_cache = {}
        for c0 in range(0, len(items), chunk):
            part = items[c0:c0 + chunk]
            samples = [it[2] for it in part]
            dA, dG = pt.batch_counts_dev(samples, ref_file, ("", ap), dev, prep_cache)
            xA = pt.prepare_batch_dev(dA, ref_file, "", ctx, prep_cache)
            xG = pt.prepare_batch_dev(dG, ref_file, ap, ctx, prep_cache)
            rows, host = wd.predict_batch_dev(be, A, G, xA, xG, rem, pt, want_host=True)
            for i, (infile, outid, _, gender, n_reads) in enumerate(part):
                results = {"results_nr": pt.ATTACHED, "results_c": rows[i]}
                for row, key in enumerate(("results_r", "results_z", "results_w")):
                    full = host[row, i]
                    results[key] = [full[off[c]:off[c + 1]] for c in range(len(off) - 1)]
                a_i = argparse.Namespace(**dict(vars(args), infile=infile, outid=outid))
                rem_i = dict(rem, args=a_i, n_reads=n_reads, gender=gender)
                if args.bed:
                    generate_output_tables(rem_i, results, ctx)
                logging.info("Finished prediction of {} -> {}".format(infile, outid))


def tool_test_batch(args, pairs):
    """`predict ... --batch LIST`: every (infile, outid) of LIST next to the positional pair, striped
    over --gpus devices (one process per GPU, no collective; BASELINE configs[4])."""
    world = max(1, int(getattr(args, "gpus", 1) or 1))
    if world == 1:
        _batch_worker(0, 1, args, pairs)
        return
    import torch.multiprocessing as mp
    mp.spawn(_batch_worker, args=(world, args, pairs), nprocs=world, join=True)


# --------------------------------------------------------------------------- CLI
def build_parser():
    parser = argparse.ArgumentParser(description="WisecondorX (MI355X hot path)")
    parser.add_argument("--loglevel", type=str, default="INFO",
                        choices=["info", "warning", "debug", "error", "critical"])
    sub = parser.add_subparsers()
    F = argparse.ArgumentDefaultsHelpFormatter

    p = sub.add_parser("convert", description="Convert and filter a aligned reads to .npz",
                       formatter_class=F)
    p.add_argument("infile", type=str, help="aligned reads input for conversion")
    p.add_argument("outfile", type=str, help="Output .npz file")
    p.add_argument("-r", "--reference", type=str, help="Fasta reference to be used during cram conversion")
    p.add_argument("--binsize", type=float, default=5e3, help="Bin size (bp)")
    p.add_argument("--normdup", action="store_true", help="Do not remove duplicates")
    p.set_defaults(func=tool_convert)

    p = sub.add_parser("newref", description="Create a new reference using healthy reference samples",
                       formatter_class=F)
    p.add_argument("infiles", type=str, nargs="+", help="Path to all reference data files")
    p.add_argument("outfile", type=str, help="Path and filename for the reference output")
    p.add_argument("--nipt", action="store_true", help="Use flag for NIPT")
    p.add_argument("--yfrac", type=float, default=None,
                   help="Use to manually set the y read fraction cutoff, which defines gender")
    p.add_argument("--plotyfrac", type=str, default=None, help="Path to yfrac .png plot")
    p.add_argument("--refsize", type=int, default=300, help="Amount of reference locations per target")
    p.add_argument("--binsize", type=int, default=1e5,
                   help="Scale samples to this binsize, multiples of existing binsize only")
    p.add_argument("--cpus", type=int, default=1, help="Accepted for compatibility (ignored)")
    p.add_argument("--gpus", type=int, default=1, help="Number of MI355X devices to split the rows over")
    p.add_argument("--aligned-masks", action="store_true",
                   help="Keep the autosomal part of the mask fixed in the gonosomal passes. Default "
                        "(like upstream WisecondorX): their PCA-distance filter may drop autosomal bins "
                        "the autosomal reference still holds -- no predict can use such a reference, so "
                        "newref then stops with an error instead of writing it")
    p.add_argument("--reference-mask-skew", action="store_true",
                   help="write the reference even when the gonosomal passes dropped autosomal bins, "
                        "exactly as upstream does (its predict then fails on it; ours refuses it)")
    p.set_defaults(func=tool_newref)

    p = sub.add_parser("gender", description="Returns the gender of a .npz resulting from convert",
                       formatter_class=F)
    p.add_argument("infile", type=str, help=".npz input file")
    p.add_argument("reference", type=str, help="Reference .npz, as previously created with newref")
    p.set_defaults(func=output_gender)

    p = sub.add_parser("predict", description="Find copy number aberrations", formatter_class=F)
    p.add_argument("infile", type=str, help=".npz input file")
    p.add_argument("reference", type=str, help="Reference .npz, as previously created with newref")
    p.add_argument("outid", type=str, help="Basename (w/o extension) of output files")
    p.add_argument("--minrefbins", type=int, default=150,
                   help="Minimum amount of sensible reference bins per target bin.")
    p.add_argument("--maskrepeats", type=int, default=5, help="Number of masking cycles.")
    p.add_argument("--alpha", type=float, default=1e-4, help="p-value cut-off for calling a CBS breakpoint.")
    p.add_argument("--zscore", type=float, default=5, help="z-score cut-off for aberration calling.")
    p.add_argument("--beta", type=float, default=None, help="ratio cut-off parameter (0,1]")
    p.add_argument("--blacklist", type=str, default=None, help="Blacklist .bed that masks regions in output")
    p.add_argument("--gender", type=str, choices=["F", "M"], help="Force the gender")
    p.add_argument("--ylim", type=str, default="def", help="y-axis limits for plotting")
    p.add_argument("--bed", action="store_true", help="Outputs tab-delimited .bed files")
    p.add_argument("--plot", action="store_true", help="Outputs .png plots (needs R; skipped)")
    p.add_argument("--cairo", action="store_true", help="Uses cairo bitmap type for plotting.")
    p.add_argument("--add-plot-title", action="store_true", help="Add the output name as plot title")
    p.add_argument("--seed", type=int, default=None, help="Seed for segmentation algorithm")
    p.add_argument("--batch", type=str, default=None,
                   help="Text file with further 'infile outid' pairs, one per line: all samples are "
                        "predicted device-resident in batches (the positional infile / outid is the first)")
    p.add_argument("--batch-size", type=int, default=96, help="Samples per device batch (--batch)")
    p.add_argument("--gpus", type=int, default=1,
                   help="--batch: stripe the samples over this many MI355X devices (one process each)")
    p.add_argument("--regions", type=str, default=None, help="Regions .bed to summarise")
    p.set_defaults(func=tool_test)
    return parser


def main(argv=None):
    warnings.filterwarnings("ignore")
    parser = build_parser()
    args = parser.parse_args(sys.argv[1:] if argv is None else argv)
    logging.basicConfig(format="[%(levelname)s - %(asctime)s]: %(message)s",
                        datefmt="%Y-%m-%d %H:%M:%S",
                        level=getattr(logging, args.loglevel.upper(), None))
    logging.debug("args are: {}".format(args))
    if not hasattr(args, "func"):
        parser.print_help()
        return
    args.func(args)


if __name__ == "__main__":
    # the command line never touches torch: skip _lib's torch preload (1 s of import time).  Only
    # here -- a process that calls main() as a function may use torch later, and then torch's
    # bundled HIP runtime has to be the first one loaded (see _lib.load).
    import os
    if "torch" not in sys.modules:
        os.environ.setdefault("WCX_NO_TORCH_PRELOAD", "1")
    main()
