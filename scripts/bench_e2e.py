#!/usr/bin/env python3
"""End-to-end wall-clock of the CLI at BASELINE.json's problem size: synthetic sample .npz files ->
`newref` (load, gender model, masks, PCA, A/F/M searches + null ratios, reference .npz written) ->
`predict` of one sample (load reference, normalise A + gonosomes, post-process, CBS, segment z,
_bins/_segments/_aberrations/_statistics tables).  This is the literal "newref+predict
wall-clock @15kb bins" of BASELINE.json; bench.py times the device-resident hot path only.
Prints one JSON line with the phase breakdown."""
import argparse
import json
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Phases:
    def __init__(self):
        self.t = {}
        self.first = {}          # wall-clock of the first entry of each phase

    def wrap(self, module, name, key):
        fn = getattr(module, name)

        def timed(*a, **k):
            t0 = time.perf_counter()
            self.first.setdefault(key, t0)
            try:
                return fn(*a, **k)
            finally:
                self.t[key] = self.t.get(key, 0.0) + time.perf_counter() - t0
        setattr(module, name, timed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--binsize", type=int, default=15000)
    ap.add_argument("--samples", type=int, default=100)
    ap.add_argument("--refsize", type=int, default=300)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workdir", default="/tmp/wcx_e2e")
    ap.add_argument("--profile", action="store_true", help="cProfile of the newref call -> stderr")
    a = ap.parse_args()

    os.environ.setdefault("WCX_NO_TORCH_PRELOAD", "1")     # like `python -m wisecondorx_amd.main`
    from wisecondorx_amd import main as cli, newref_tools, npz_io, predict_tools, prep, synth
    from wisecondorx_amd import predict_output

    shutil.rmtree(a.workdir, ignore_errors=True)
    os.makedirs(a.workdir)
    t0 = time.perf_counter()
    co = synth.Cohort(a.binsize, female_y=0.1)
    samples, genders = co.cohort(a.samples)
    files = []
    for i, s in enumerate(samples):
        f = os.path.join(a.workdir, "s{:03d}.npz".format(i))
        npz_io.save_sample(f, s, a.binsize)
        files.append(f)
    test_file = os.path.join(a.workdir, "test.npz")
    npz_io.save_sample(test_file, co.sample(9001, "F", cnv=[(3, 500, 500 + int(3e7 / a.binsize), 1.5)]),
                       a.binsize)
    t_synth = time.perf_counter() - t0

    ph = Phases()
    ph.wrap(npz_io, "load_sample", "load_samples_thread_sum")   # 8 loader threads: summed, not wall
    ph.wrap(prep, "get_mask", "masks")
    ph.wrap(prep, "prepare", "prep_pca")
    ph.wrap(newref_tools, "get_reference_parts", "gpu_search_nullratios")
    ph.wrap(npz_io, "save_npz", "write_reference")
    ph.wrap(cli, "train_gender_model", "gender_model")
    from wisecondorx_amd import ref_qc
    ph.wrap(prep.DeviceCounts, "__init__", "counts_to_device")
    ph.wrap(prep.DeviceCounts, "get_mask", "masks")
    ph.wrap(prep, "prepare_dev", "prep_pca")
    ph.wrap(newref_tools, "get_reference_dev", "gpu_search_nullratios")
    ph.wrap(ref_qc, "qc_reference", "reference_qc")
    ph.wrap(npz_io.NpzWriter, "close", "writer_close")
    ph.wrap(npz_io.os, "fsync", "fsync_thread_sum")            # (worker threads + close(): summed)

    ref_file = os.path.join(a.workdir, "ref.npz")
    import random
    random.seed(1)
    newref_argv = ["--loglevel", "warning", "newref"] + files + [
        ref_file, "--binsize", str(a.binsize), "--refsize", str(a.refsize), "--yfrac", "0.004",
        "--gpus", str(a.gpus), "--aligned-masks"]   # (the synthetic cohort's F / M filters drop autosomal bins)
    t0 = time.perf_counter()
    if a.profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.runcall(cli.main, newref_argv)
        pstats.Stats(pr, stream=sys.stderr).sort_stats(os.environ.get("WCX_PROF_SORT", "cumulative")).print_stats(45)
    else:
        cli.main(newref_argv)
    t_newref = time.perf_counter() - t0
    newref_phases = dict(ph.t)
    if "gender_model" in ph.first:       # wall-clock of the 8-thread import (up to the gender model)
        newref_phases["load_samples_wall"] = ph.first["gender_model"] - t0

    ph.t = {}
    ph.wrap(npz_io, "load_reference", "load_reference")
    ph.wrap(predict_tools, "normalize", "normalize")
    ph.wrap(predict_tools, "exec_cbs", "cbs_and_segment_z")
    ph.wrap(predict_tools, "get_post_processed_result", "post_process")
    ph.wrap(predict_tools, "attach_null_matrix", "null_matrix_upload")
    ph.wrap(predict_output, "generate_output_tables", "tables")
    outid = os.path.join(a.workdir, "out")
    t0 = time.perf_counter()
    predict_error = None
    predict_argv = ["--loglevel", "warning", "predict", test_file, ref_file, outid, "--bed", "--seed", "1"]
    try:
        if a.profile:
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.runcall(cli.main, predict_argv)
            pstats.Stats(pr, stream=sys.stderr).sort_stats(os.environ.get("WCX_PROF_SORT", "cumulative")).print_stats(45)
        else:
            cli.main(predict_argv)
    except SystemExit as e:     # e.g. a reference whose gonosomal pass dropped autosomal bins
        predict_error = "predict exited ({})".format(e.code)
    t_predict = time.perf_counter() - t0
    seg = ab = None
    if predict_error is None:
        seg = sum(1 for _ in open(outid + "_segments.bed")) - 1
        ab = sum(1 for _ in open(outid + "_aberrations.bed")) - 1
    out = {
        "workload": "CLI newref ({} samples, {} bp bins, refsize {}) + predict of 1 sample".format(
            a.samples, a.binsize, a.refsize),
        "n_gpus": a.gpus, "newref_s": t_newref, "predict_s": t_predict,
        "newref_plus_predict_s": t_newref + t_predict,
        "newref_phases_s": newref_phases, "predict_phases_s": dict(ph.t),
        "synth_and_write_samples_s": t_synth,
        "reference_npz_bytes": os.path.getsize(ref_file), "segments": seg, "aberrations": ab,
    }
    if predict_error:
        out["predict_error"] = predict_error
    print(json.dumps(out))
    shutil.rmtree(a.workdir, ignore_errors=True)


if __name__ == "__main__":
    main()
