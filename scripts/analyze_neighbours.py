#!/usr/bin/env python3
"""dev helper (GPU box): structure of the reference-bin sets of the bench workload -- how concentrated
the candidates are (hubs) and how much the sets of nearby targets overlap under different target
orders.  Input to the design of the refine kernel's locality."""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from wisecondorx_amd import newref_tools as nt

S = int(sys.argv[1]) if len(sys.argv) > 1 else 500
co, p, test = bench.make_workload(15000, S)
X = p["X"]
cum = np.asarray(p["masked_bins_per_chr_cum"], dtype=np.int64)
B = int(cum[-1])
t = time.time()
idx, dist = nt.get_ref_for_rows(X, cum, 300, 0, B, mode=2)
print("search", time.time() - t, "s; B", B, flush=True)
# own-chromosome-excluded index -> global row
mb = np.diff(np.concatenate(([0], cum)))
own = np.repeat(mb, mb)
cs = np.repeat(np.concatenate(([0], cum[:-1])), mb)
g = idx + (idx >= cs[:, None]) * own[:, None]
cnt = np.bincount(g.ravel(), minlength=B)
order = np.argsort(-cnt)
cs_ = np.cumsum(cnt[order]) / cnt.sum()
for f in (0.001, 0.01, 0.02, 0.05, 0.1, 0.2, 0.5):
    print("top %5.1f %% of the candidates cover %.3f of the pairs" % (100 * f, cs_[int(f * B) - 1]))
print("candidates used at all: %d of %d" % ((cnt > 0).sum(), B))

def block_union(order_rows, blk):
    """mean number of distinct candidates in blocks of `blk` consecutive targets of the order"""
    u = []
    for a in range(0, len(order_rows) - blk, max(blk, len(order_rows) // 200)):
        u.append(len(np.unique(g[order_rows[a:a + blk]])))
    return float(np.mean(u))

norm = np.sqrt(((np.asarray(X) - 1.0) ** 2).sum(axis=1))
orders = {"row order": np.arange(B), "by norm": np.argsort(norm), "by first neighbour": np.argsort(g[:, 0], kind="stable"),
          "by mean neighbour rank": np.argsort(np.argsort(order)[g].mean(axis=1))}
for name, o in orders.items():
    print("%-24s distinct candidates per block of 16 / 64 / 256 / 1024 targets: %.0f %.0f %.0f %.0f" % (
        name, block_union(o, 16), block_union(o, 64), block_union(o, 256), block_union(o, 1024)), flush=True)
# mutual pairs
key = (np.repeat(np.arange(B), 300).astype(np.int64) << 20 | 0)  # placeholder to keep memory low
pairs = np.stack((np.repeat(np.arange(B), 300), g.ravel()), axis=1)
a = np.minimum(pairs[:, 0], pairs[:, 1]).astype(np.int64) * B + np.maximum(pairs[:, 0], pairs[:, 1])
u, c = np.unique(a, return_counts=True)
print("pairs %d, unordered distinct %d, mutual fraction of pairs %.3f" % (len(a), len(u), 2 * (c == 2).sum() / len(a)))
