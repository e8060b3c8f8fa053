#!/usr/bin/env python3
"""dev helper: one line per bench JSON (step ms, screen/refine/null ms, appends per row, phases)."""
import json
import sys

NAMES = ["mfma", "stage", "barrier", "sign", "append", "cuts"]
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        r = d["roofline"]
        rows = d["config"]["bins"]
        line = "%-34s step %7.2f  screen %6.2f (frac %.3f) prep %.2f refine %5.2f null %5.2f norm %s  app/row %6.0f cuts/row %.2f fb %d" % (
            f.split("/")[-1], d["ms_per_step"], r["kernel_ms"], r["frac"], r["prep_ms"], r["refine_ms"],
            r["null_ratios_ms"], r.get("normalize_ms"), r["appends"] / rows, r["compactions"] / rows,
            r["fallback_rows"])
        pc = r.get("phase_cycles")
        if pc and sum(pc):
            tot = float(sum(pc))
            line += "  | " + " ".join("%s %.0f%%" % (n, 100 * c / tot) for n, c in zip(NAMES, pc))
        print(line)
    except Exception as e:
        print(f, "ERR", e)
