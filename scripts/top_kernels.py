#!/usr/bin/env python3
"""dev helper: top rows of a rocprofv3 kernel_stats.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    n = r['Name'].replace('(anonymous namespace)::', '')[:56]
    print(f"{n:56s} calls {r['Calls']:>5s} total {float(r['TotalDurationNs'])/1e6:8.2f} ms avg {float(r['AverageNs'])/1e3:9.1f} us")
