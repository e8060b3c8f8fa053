#!/usr/bin/env python3
"""Round 6, one-directional sweep: thresholds from hub counts (screen_hub1.h) against the sampled
pre-pass, per consumer -- the A pass of configs[2] (15 kb x 100) and configs[1] (100 kb x 100), the F / M
passes of the 500-sample cohort -- over hub fraction and trial count.
Device times from the context's timers (HIP events), counters from wcx_last_topk_stats; every variant's
tables must equal the first one's bit for bit.  Writes gpurun_out/sweep_hub1.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from wisecondorx_amd import _lib, newref_tools as nt  # noqa: E402

VARIANTS = [
    ("sampled", {"WCX_SCREEN_HUB": "0"}),
    ("hub", {}),
    ("hub f8", {"WCX_HUB_FRAC": "8"}),
    ("hub f16", {"WCX_HUB_FRAC": "16"}),
    ("hub f24", {"WCX_HUB_FRAC": "24"}),
    ("hub f32", {"WCX_HUB_FRAC": "32"}),
    ("hub f64", {"WCX_HUB_FRAC": "64"}),
]
KEYS = ("WCX_SCREEN_HUB", "WCX_HUB_FRAC", "WCX_HUB1_TRIALS", "WCX_HUB_N1", "WCX_SCREEN_SEGMENTS_SMALL",
        "WCX_SCREEN_CHUNK_KB_SMALL", "WCX_SCREEN_TILE", "WCX_SCREEN_SEGMENTS")


def run(tag, X, cum, k, r0, r1, variants, reps=4):
    ctx = _lib.default_context()
    out, first = [], None
    for name, env in variants:
        for k_ in KEYS:
            os.environ.pop(k_, None)
        os.environ.update(env)
        ms = {}
        for _ in range(reps):
            idx, dist = nt.get_ref_for_rows(X, cum, k, r0, r1, mode=0)
            for t in ("topk", "topk_prep", "topk_screen", "topk_pre", "topk_refine"):
                ms.setdefault(t, []).append(ctx.kernel_ms(t))
        st = ctx.topk_stats()
        if first is None:
            first = (idx, dist)
        same = bool(np.array_equal(idx, first[0]) and np.array_equal(dist, first[1]))
        row = {"workload": tag, "variant": name, "same_bits": same, "rows": st["rows"],
               "appends_per_row": st["appends"] / max(1, st["rows"]), "cuts_per_row": st["compactions"] / max(1, st["rows"]),
               "refined_per_row": st["refined"] / max(1, st["rows"]), "fallback_rows": st["fallback_rows"],
               "rows_without_estimate": st["hub_rows_without_estimate"],
               "mean_trial": st["hub_trial_sum"] / max(1, st["rows"])}
        row.update({t + "_ms": float(np.min(v)) for t, v in ms.items()})
        out.append(row)
        print("{:12s} {:14s} screen {:6.2f} (pre {:5.2f}) refine {:5.2f} topk {:6.2f}  app/row {:6.0f} cuts/row {:4.2f} "
              "refined/row {:4.0f} fb {} noest {} trial {:.2f} same {}".format(
                  tag, name, row["topk_screen_ms"], row["topk_pre_ms"], row["topk_refine_ms"], row["topk_ms"],
                  row["appends_per_row"], row["cuts_per_row"], row["refined_per_row"], row["fallback_rows"],
                  row["rows_without_estimate"], row["mean_trial"], same), flush=True)
    for k_ in KEYS:
        os.environ.pop(k_, None)
    return out


def main():
    which = sys.argv[1:] or ["100kb", "S100", "FM"]
    res = []
    if "100kb" in which:
        p = bench.make_full_workload(100000, 100)[1]["A"]
        cum = [int(v) for v in p["masked_bins_per_chr_cum"]]
        res += run("100kb x100 A", p["X"], cum, 300, 0, cum[-1], VARIANTS[:3] + [
            ("hub t4", {"WCX_HUB1_TRIALS": "4"}), ("hub tt2", {"WCX_SCREEN_TILE": "2,2,4,2,3"}),
            ("hub seg2", {"WCX_SCREEN_SEGMENTS": "2"}), ("hub seg3", {"WCX_SCREEN_SEGMENTS": "3"}),
            ("hub tt2 seg6", {"WCX_SCREEN_TILE": "2,2,4,2,3", "WCX_SCREEN_SEGMENTS": "6"}),
            ("hub tt2 seg3", {"WCX_SCREEN_TILE": "2,2,4,2,3", "WCX_SCREEN_SEGMENTS": "3"})])
    if "S100" in which:
        p = bench.make_full_workload(15000, 100)[1]["A"]
        cum = [int(v) for v in p["masked_bins_per_chr_cum"]]
        res += run("15kb x100 A", p["X"], cum, 300, 0, cum[-1], VARIANTS + [
            ("hub t8", {"WCX_HUB1_TRIALS": "8"}), ("hub tt2", {"WCX_SCREEN_TILE": "2,2,4,2,3"}),
            ("sampled tt2", {"WCX_SCREEN_HUB": "0", "WCX_SCREEN_TILE": "2,2,4,2,3"})])
    if "FM" in which:
        passes = bench.make_full_workload(15000, 500)[1]
        for tag in ("F", "M"):
            p = passes[tag]
            cum = [int(v) for v in p["masked_bins_per_chr_cum"]]
            res += run("15kb x500 " + tag, p["X"], cum, 300, cum[21], cum[-1],
                       VARIANTS[:2] + [VARIANTS[3], VARIANTS[6], ("hub t4", {"WCX_HUB1_TRIALS": "4"}),
                                       ("hub seg2", {"WCX_SCREEN_SEGMENTS_SMALL": "2"}),
                                       ("hub seg5", {"WCX_SCREEN_SEGMENTS_SMALL": "5"}),
                                       ("hub seg6", {"WCX_SCREEN_SEGMENTS_SMALL": "6"}),
                                       ("hub seg8", {"WCX_SCREEN_SEGMENTS_SMALL": "8"}),
                                       ("hub seg8 c48", {"WCX_SCREEN_SEGMENTS_SMALL": "8", "WCX_SCREEN_CHUNK_KB_SMALL": "49152"}),
                                       ("hub c48", {"WCX_SCREEN_CHUNK_KB_SMALL": "49152"}),
                                       ("hub c12", {"WCX_SCREEN_CHUNK_KB_SMALL": "12288"})])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "sweep_hub1.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
