#!/usr/bin/env python3
"""Condense the rocprofv3 output of scripts_prof.sh (gpurun_out/prof_<tag>/) into the small files
kept under profiles/<round>/:
  kernel_stats_<tag>.csv     rocprofv3 --kernel-trace --stats summary (as produced)
  pmc_<tag>_pmc<N>.csv       per-kernel sums of every counter of PMC pass N (+ dispatch count)
  screen_traffic.json        per-sweep HBM bytes and SQ breakdown of the dominant kernel (k_screen),
                             read by bench.py for roofline.traffic
usage: python scripts/summarize_prof.py <tag> [round_dir=profiles/r01] [steps=3]"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    rdir = os.path.join(ROOT, sys.argv[2] if len(sys.argv) > 2 else "profiles/r01")
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3      # bench steps + warmup in the profiled run
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    os.makedirs(rdir, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"),
                os.path.join(rdir, "kernel_stats_{}.csv".format(tag)))
    per_pass = {}
    for n in (1, 2, 3, 4):
        f = os.path.join(src, "pmc{}".format(n), "p_counter_collection.csv")
        if not os.path.exists(f):
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"][:40]
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[name].add(r["Dispatch_Id"])
        counters = sorted({c for k in agg for c in agg[k]})
        with open(os.path.join(rdir, "pmc_{}_pmc{}.csv".format(tag, n)), "w") as out:
            out.write("Kernel," + ",".join(counters) + ",dispatches\n")
            for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
                out.write(k.replace(",", ";") + "," + ",".join(str(agg[k].get(c, 0.0)) for c in counters)
                          + "," + str(len(disp[k])) + "\n")
        per_pass[n] = {k: (dict(v), len(disp[k])) for k, v in agg.items()}

    def screen(n):
        for k, (v, d) in per_pass.get(n, {}).items():
            if "k_screenILi" in k and "prep" not in k:
                return v, d
        return {}, 0
    p1, d1 = screen(1)
    p2, _ = screen(2)
    p3, d3 = screen(3)
    p4, d4 = screen(4)
    if d1:
        launches = d1 / steps
        tj = {
            "kernel": "k_screen<7,2> (15 kb, B=182179, S=100, k=300)",
            "launches_per_sweep": launches,
            "fetch_bytes_per_sweep_raw": p3.get("FETCH_SIZE", 0.0) * 1024 / steps if d3 else None,
            "write_bytes_per_sweep": p4.get("WRITE_SIZE", 0.0) * 1024 / steps if d4 else None,
            "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB) in separate passes, summed over the "
                    "chunk launches of one screen sweep; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                    "(gfx950 counts 64-B units of 128-B requests); WRITE_SIZE as reported",
            "mfma_busy_cycles_per_sweep": p1.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / steps,
            "wave_quadcycles_per_sweep": p1.get("SQ_WAVE_CYCLES", 0.0) / steps,
            "wait_any_frac": p1.get("SQ_WAIT_ANY", 0.0) / max(p1.get("SQ_WAVE_CYCLES", 1.0), 1.0),
            "wait_inst_frac": p1.get("SQ_WAIT_INST_ANY", 0.0) / max(p1.get("SQ_WAVE_CYCLES", 1.0), 1.0),
            "active_frac": p1.get("SQ_ACTIVE_INST_ANY", 0.0) / max(p1.get("SQ_WAVE_CYCLES", 1.0), 1.0),
            "valu_insts_per_sweep": p2.get("SQ_INSTS_VALU", 0.0) / steps,
            "salu_insts_per_sweep": p2.get("SQ_INSTS_SALU", 0.0) / steps,
            "lds_insts_per_sweep": p2.get("SQ_INSTS_LDS", 0.0) / steps,
            "vmem_wr_insts_per_sweep": p2.get("SQ_INSTS_VMEM_WR", 0.0) / steps,
        }
        if tj["fetch_bytes_per_sweep_raw"] is not None:
            tj["fetch_bytes_per_sweep_corrected_x2"] = 2.0 * tj["fetch_bytes_per_sweep_raw"]
        json.dump(tj, open(os.path.join(rdir, "screen_traffic.json"), "w"), indent=1)
        print(json.dumps(tj, indent=1))


if __name__ == "__main__":
    main()
