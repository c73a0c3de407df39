#!/usr/bin/env python3
"""Per (kernel, grid size) durations from a rocprofv3 rocpd database: inside a training step the grid size identifies
the GEMM shape, so this gives IN-STEP time per shape to set against tools/bench_gemm.py's isolated numbers.
usage: rocpd_per_grid.py db [name-filter]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else "gemm"
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
gx = [k for k in ("grid_x", "grid_size_x", "grid_size") if k in cols]
wx = [k for k in ("workgroup_x", "workgroup_size_x", "workgroup_size") if k in cols]
if not gx:
    print("columns:", cols)
    sys.exit(1)
rows = c.execute(f"select name, {gx[0]}, {wx[0] if wx else 0}, count(*), avg(end-start), min(end-start), max(end-start) from kernels "
                 f"group by name, {gx[0]} order by 4*5 desc").fetchall()
for name, g, w, n, avg, lo, hi in rows:
    if flt not in name:
        continue
    short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
    short = re.sub(r"^void \(anonymous namespace\)::", "", short)[:40]
    print(f"{short:42s} grid {g:>8} wg {w:>4} n {n:5d}  avg {avg / 1e3:8.1f} us  min {lo / 1e3:8.1f}  max {hi / 1e3:8.1f}")
