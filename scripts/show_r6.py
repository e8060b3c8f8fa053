#!/usr/bin/env python3
"""dev helper: the round-6 fields of a bench JSON line."""
import json, sys
for f in sys.argv[1:]:
    p = json.loads(open(f).read().strip().split("\n")[-1])
    r = p["roofline"]
    print(f, "step %.2f frac %.3f screen %.2f pre %.2f refine %.2f null %.2f" % (p["ms_per_step"], r["frac"], r["kernel_ms"], r.get("pre_ms", -1), r["refine_ms"], r["null_ratios_ms"]))
    print(" verified", {k: v for k, v in (p.get("verified") or {}).items() if k in ("rows_all_passes", "mismatches_all_passes")})
    rc = p.get("rccl_world1")
    if rc:
        print(" rccl_world1 ok", rc["ok"], rc.get("differences"), rc.get("error"), {k: (v["calls"], round(v["ms"], 2)) for k, v in rc.get("collectives", {}).items()})
    s = p.get("secondary")
    if s:
        sr = s["roofline"]
        print(" S100 step %.2f frac %.3f screen %.2f pre %.2f appends %d" % (s["ms_per_step"], sr["frac"], sr["kernel_ms"], sr.get("pre_ms", -1), sr["appends"]), (s.get("verified") or {}).get("mismatches_all_passes"))
    c = p.get("config2_100kb")
    if c:
        print(" 100kb step %.2f screen %.2f pre %s refine %.2f null %.2f appends %s" % (c["ms_per_step"], c["screen_ms"], c.get("pre_ms"), c["refine_ms"], c["null_ratios_ms"], c.get("appends")), (c.get("verified") or {}).get("mismatches_all_passes"), (c.get("verified") or {}).get("rows_all_passes"))
    c = p.get("config5")
    if c:
        print(" cfg5 batch %.4f prep %.4f" % (c["batch_s"], c["prep_s"]), {k: round(v, 2) for k, v in c["kernel_ms"].items()}, {k: v for k, v in (c.get("verified") or {}).items() if k not in ("what",)})
    for k in ("e2e_cli", "e2e_cli_100kb"):
        if k in p:
            print(" ", k, round(p[k].get("newref_s", -1), 3), round(p[k].get("predict_s", -1), 3))
    g = r.get("gonosomal_passes", {})
    for t in g:
        print("  ", t, {k: round(v, 2) for k, v in g[t].items() if k.endswith("ms")})
    hk = p.get("hbm_kernels", {})
    for k, v in hk.items():
        print("   hbm", k[:40], round(v["ms"], 3), round(v["frac_of_hbm_peak"], 3))
