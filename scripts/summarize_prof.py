#!/usr/bin/env python3
"""Condense gpurun_out/prof_<round> (scripts/measure_traffic.sh) into profiles/<round>/ (round = $WCX_PROF_ROUND, default r06):
  kernel_stats_S<S>.csv     rocprofv3 --kernel-trace --stats summary of the bench command
  pmc_S<S>.csv              per-kernel sums of every counter (all PMC passes) + dispatch counts
  screen_traffic.json       per-sweep HBM bytes and SQ breakdown of k_screen, keyed by workload, with
                            the hash of the kernel sources (bench.py only reports roofline.traffic
                            while that hash matches)
usage: python scripts/summarize_prof.py [sweeps_in_profiled_run=2]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ROUND = os.environ.get("WCX_PROF_ROUND", "r06")
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + ROUND)
DST = os.path.join(ROOT, "profiles", ROUND)


def short_name(full):
    """`void (anonymous namespace)::k_foo<5>(args...)` -> `k_foo<5>`; library kernels keep the tail of
    their qualified name."""
    n = full.replace("(anonymous namespace)::", "")
    if n.startswith("void "):
        n = n[5:]
    depth, cut = 0, len(n)
    for i, ch in enumerate(n):                      # argument list = first '(' outside <...>
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    n = n[:cut].strip()
    return n if len(n) <= 70 else "..." + n[-67:]


def main():
    sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 2      # warmup 1 + steps 1
    import bench
    os.makedirs(DST, exist_ok=True)
    out = {"kernel_sha": bench.screen_source_sha(), "workloads": {},
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB) in separate passes, summed over the "
                   "chunk launches of one screen sweep; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                   "(gfx950 counts 64-B units of 128-B requests); WRITE_SIZE as reported"}
    for S in (500, 100):
        st = glob.glob(os.path.join(SRC, "trace_S%d" % S, "**", "t_kernel_stats.csv"), recursive=True)
        if st:
            shutil.copy(st[0], os.path.join(DST, "kernel_stats_S%d.csv" % S))
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for n in (1, 2, 3, 4):
            for f in glob.glob(os.path.join(SRC, "pmc%d_S%d" % (n, S), "**", "p_counter_collection.csv"),
                               recursive=True):
                for r in csv.DictReader(open(f)):
                    name = short_name(r["Kernel_Name"])
                    if "k_screen_sym<" in r["Kernel_Name"]:
                        name = "k_screen_sym"       # the symmetric sweep of the autosomal pass (one launch)
                    elif "k_screen_count<" in r["Kernel_Name"]:
                        name = "k_screen_count"     # its thresholds: counts over the hub region
                    elif "k_screen_hub1<" in r["Kernel_Name"]:
                        import re
                        nk = int(re.search(r"k_screen_hub1<(\d+)", r["Kernel_Name"]).group(1))
                        # thresholds of the one-directional sweep (round 6): the A pass's at S = 100
                        name = "k_screen_hub1" if nk == (32 if S == 500 else 7) else "k_screen_hub1 (gonosomal passes)"
                    elif "k_screen<" in r["Kernel_Name"] and "prep" not in r["Kernel_Name"]:
                        # the step runs three passes: the autosomal one (all S samples: the largest
                        # NK of the run) is the dominant kernel, the two gonosomal ones (S / 2
                        # samples) are kept apart
                        import re
                        nk = int(re.search(r"k_screen<(\d+)", r["Kernel_Name"]).group(1))
                        name = "k_screen" if nk == (32 if S == 500 else 7) else "k_screen (gonosomal passes)"
                    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
                    disp[(name, n)].add(r["Dispatch_Id"])
        if not agg:
            continue
        counters = sorted({c for k in agg for c in agg[k]})
        with open(os.path.join(DST, "pmc_S%d.csv" % S), "w") as fh:
            fh.write("Kernel," + ",".join(counters) + ",dispatches\n")
            for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0.0)):
                fh.write(k.replace(",", ";") + "," + ",".join(str(agg[k].get(c, 0.0)) for c in counters)
                         + "," + str(len(disp[(k, 1)])) + "\n")
        # the autosomal sweep = (symmetric path) the pre-pass k_screen + k_screen_sym, or k_screen alone
        sym = "k_screen_sym" in agg
        hub = "k_screen_count" in agg
        p = collections.defaultdict(float)
        # (with hub-count thresholds the sampled pre-pass k_screen only runs behind a gate that stays
        #  closed: its launches return at once and are counted with the sweep all the same)
        for part in (("k_screen", "k_screen_count", "k_screen_sym") if sym else ("k_screen", "k_screen_hub1")):
            for c_, v_ in agg[part].items():
                p[c_] += v_
        wave = p.get("SQ_WAVE_CYCLES", 0.0)
        out["workloads"]["S%d" % S] = {
            "kernels": ("k_screen_count (hub-count thresholds) + k_screen_sym (+ the gated second attempt's "
                        "empty launches)" if hub else "k_screen (sampled pre-pass) + k_screen_sym") if sym else
                       ("k_screen_hub1 (hub-count thresholds) + k_screen" if "k_screen_hub1" in agg else "k_screen"),
            "launches_per_sweep": (len(disp[("k_screen", 1)]) + len(disp[("k_screen_sym", 1)]) +
                                   len(disp[("k_screen_count", 1)])) / sweeps,
            "fetch_bytes_per_sweep_raw": p.get("FETCH_SIZE", 0.0) * 1024 / sweeps,
            "fetch_bytes_per_sweep_corrected_x2": 2 * p.get("FETCH_SIZE", 0.0) * 1024 / sweeps,
            "write_bytes_per_sweep": p.get("WRITE_SIZE", 0.0) * 1024 / sweeps,
            "mfma_busy_cycles_per_sweep": p.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / sweeps,
            "wave_quadcycles_per_sweep": wave / sweeps,
            "wait_any_frac": p.get("SQ_WAIT_ANY", 0.0) / wave if wave else None,
            "wait_inst_frac": p.get("SQ_WAIT_INST_ANY", 0.0) / wave if wave else None,
            "active_frac": p.get("SQ_ACTIVE_INST_ANY", 0.0) / wave if wave else None,
            "insts_per_sweep": {c: p.get(c, 0.0) / sweeps for c in counters if c.startswith("SQ_INSTS_")},
            "gui_active_cycles_per_sweep_all_xcc": p.get("GRBM_GUI_ACTIVE", 0.0) / sweeps,
        }
    json.dump(out, open(os.path.join(DST, "screen_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
