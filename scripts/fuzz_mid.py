#!/usr/bin/env python3
"""One-off randomised parity sweep like tests/test_gpu_fuzz.py on MID-SIZED problems (8 k .. 40 k rows:
the range where small-K searches now run a 1/8 sampled pre-pass) against the C oracle.
usage: fuzz_mid.py [first_seed] [n]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_gpu_fuzz as F
    from oracle import c_oracle as CO
    from oracle import wcx_oracle as O
    from wisecondorx_amd import _lib, newref_tools as nt
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    bad = 0
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        X, cum, k, s, e = F._case(seed)
        rep = int(rng.integers(3, 6))                   # rows x 3 .. 5: every chromosome repeated, new noise
        Xb = np.concatenate([X * (1.0 + 0.02 * rng.standard_normal(X.shape)) for _ in range(rep)], axis=0)
        mb = np.diff([0] + list(cum))
        cumb = np.cumsum(np.tile(mb, rep)[:24] if len(mb) * rep > 24 else np.tile(mb, rep)).tolist()
        Xb = np.asfortranarray(Xb[:cumb[-1]])
        B = cumb[-1]
        s = int(rng.integers(0, B - 400))
        e = int(min(B, s + rng.integers(100, 700)))
        oi, od = CO.get_reference_rows(np.ascontiguousarray(np.asarray(Xb).T), cumb, s, e, k)
        ok = True
        for mode in (2, 0):
            idx, dist = nt.get_ref_for_rows(Xb, cumb, k, s, e, mode=mode)
            ok = ok and np.array_equal(idx, oi) and np.array_equal(dist, od)
        st = _lib.default_context().topk_stats()
        print("seed %d B %d S %d k %d rows %d: %s (fallback rows %d, refined %d)" %
              (seed, B, Xb.shape[1], k, e - s, "ok" if ok else "MISMATCH", st["fallback_rows"], st["refined"]))
        bad += 0 if ok else 1
    print("mismatching cases:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
