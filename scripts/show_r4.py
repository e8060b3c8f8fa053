#!/usr/bin/env python3
"""dev helper: one line per bench JSON (step, screen split, list statistics)."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read())
        r = d["roofline"]
        print(f[-22:], "step", round(d["ms_per_step"], 2), "screen", round(r["kernel_ms"], 2),
              "pre", round(r.get("pre_ms", -1), 2), "refine", round(r.get("refine_ms", -1), 2),
              "nr", round(r.get("null_ratios_ms", -1), 2), "app", r.get("appends"),
              "row", r.get("sym_row_appends"), "gates", r.get("sym_gates"), r.get("sym_counts"),
              "refined", r.get("refined_pairs"), "fb", r.get("fallback_rows"), d.get("verified"))
    except Exception as e:  # noqa: BLE001
        print("ERR", f, e, open(f).read()[:300])
