#!/usr/bin/env python3
"""Per-kernel summary (the `--stats` table) from a rocprofv3 rocpd database, as CSV + a short listing.

    python tools/rocpd_kernel_stats.py gpurun_out/prof_c/runc_results.db profiles/r01_c_kernel_stats_final.csv [steps]
"""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100 * r[2] / tot, 3), r[4], r[5]])
    for r in rows[:24]:
        name = r[0].split("(")[0][-58:]
        print("%-58s %6d %8.3f ms/step  avg %8.1f us %5.1f%%" % (name, r[1], r[2] / steps / 1e6, r[3] / 1e3,
                                                                 100 * r[2] / tot))
    print("all kernels: %.3f ms/step" % (tot / steps / 1e6))


if __name__ == "__main__":
    main()
