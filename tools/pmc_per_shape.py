#!/usr/bin/env python3
"""Run-length view of one PMC counter from a rocprofv3 rocpd database: consecutive dispatches of the same kernel with
similar values are one group -- with tools/bench_gemm.py (shape after shape) this gives FETCH_SIZE / WRITE_SIZE per SHAPE.
usage: pmc_per_shape.py db COUNTER [name-filter]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
flt = sys.argv[3] if len(sys.argv) > 3 else "gemm"
rows = c.execute("select name, counter_value from pmc_events where counter_name = ? order by dispatch_id", (sys.argv[2],)).fetchall()
groups = []
for name, v in rows:
    if flt not in name:
        continue
    short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:44]
    g = groups[-1] if groups else None
    if g and g[0] == short and abs(v - g[2] / g[1]) <= 0.15 * (g[2] / g[1]):
        g[1] += 1
        g[2] += v
    else:
        groups.append([short, 1, v])
for name, n, s in groups:
    print(f"{name:46s} n {n:4d}  avg {s / n * 1024 / 1e6:9.1f} MB (raw counter)")
