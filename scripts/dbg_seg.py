import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
from wisecondorx_amd import _lib, newref_tools as nt
from wisecondorx_amd.synth import bins_per_chr, corrected_matrix
bpc = [int(b * 0.95) for b in bins_per_chr(15000)[:22]]
S, k = int(sys.argv[1]), 300
X, mbpc, cum = corrected_matrix(bpc, S, seed=45)
B = cum[-1]
s, e = nt._get_part(3, 8, B)
for seg in sys.argv[2].split(","):
    for samp in ("16", "0"):
        os.environ["WCX_SCREEN_SEGMENTS"] = seg
        os.environ["WCX_SCREEN_SAMPLE"] = samp
        pi, pd = nt.get_ref_for_rows(X, cum, k, s, e, mode=2)
        st = _lib.default_context().topk_stats()
        print("S", S, "segments", seg, "sample", samp, "fallback", st["fallback_rows"], "app/row", st["appends"] / (e - s), "cuts/row", st["compactions"] / (e - s), flush=True)
