#!/usr/bin/env python3
"""GPU timeline of the bench step from a rocprofv3 kernel trace (…_kernel_trace.csv): the last step of
the run is cut at the longest idle gaps, busy / idle time and the largest gaps between kernels are
listed with the kernels on either side.  Shows what the host orchestration leaves on the table."""
import csv
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]


def main():
    path = sys.argv[1]
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    t0 = rows[0][0]
    # steps: find the k_screen<32 (A pass) first launches as step markers
    marks = [i for i, r in enumerate(rows) if "k_row_norm" in r[2]]
    print("kernels", len(rows), "row_norm marks", len(marks))
    # a step = 3 passes = 3 k_row_norm; take the last complete step
    if len(marks) >= 6:
        a, b = marks[-6], marks[-3]
    else:
        a, b = marks[0], len(rows) - 1
    seg = rows[a:b]
    start, end = seg[0][0], max(r[1] for r in seg)
    # union of busy intervals
    busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
    gaps = []
    last_name = seg[0][2]
    for s, e, n in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, (cur_e - start) / 1e6, short(last_name), short(n)))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        if e >= cur_e:
            last_name = n
    busy += cur_e - cur_s
    print("step span %.2f ms, busy %.2f ms, idle %.2f ms in %d gaps" % ((end - start) / 1e6, busy / 1e6,
          (end - start - busy) / 1e6, len(gaps)))
    for g in sorted(gaps, reverse=True)[:25]:
        print("  gap %7.1f us at %6.2f ms  after %-45s before %s" % (g[0] / 1e3, g[1], g[2], g[3]))
    hist = [0, 0, 0, 0]
    for g in gaps:
        us = g[0] / 1e3
        hist[0 if us < 5 else 1 if us < 20 else 2 if us < 100 else 3] += g[0]
    print("idle by gap size: <5us %.2f ms, 5-20us %.2f ms, 20-100us %.2f ms, >100us %.2f ms" %
          tuple(h / 1e6 for h in hist))


if __name__ == "__main__":
    main()
