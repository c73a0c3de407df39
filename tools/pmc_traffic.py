#!/usr/bin/env python3
"""HBM-side bytes per launch of the GEMM kernel families from two rocprofv3 PMC passes (rocpd databases):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out_f -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out_w -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic.py out_f/f_results.db out_w/w_results.db profiles/rNN_pmc_traffic.json

FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC counter budget); both are KiB per dispatch.  gfx950 correction
(MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so it
is DOUBLED; WRITE_SIZE is exact.  bench.py reads kernels.<family>.hbm_bytes_per_launch from the resulting file.
"""
import json
import os
import sqlite3
import sys

FAMILIES = {"gemm_nt": "gemm_nt_", "gemm_tn": "gemm_tn_", "attention": "attn_", "layernorm": "ln_"}


def per_family(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select name, counter_value from pmc_events where counter_name = ?", (counter,)).fetchall()
    out = {}
    for fam, key in FAMILIES.items():
        vals = [v for n, v in rows if key in n]
        if vals:
            out[fam] = (len(vals), sum(vals) / len(vals))
    return out


def main():
    fetch, write = per_family(sys.argv[1], "FETCH_SIZE"), per_family(sys.argv[2], "WRITE_SIZE")
    res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over "
                     "bench.py --steps 2 --warmup 1 --no-cpu-baseline; FETCH_SIZE doubled per the gfx950 correction "
                     "(tools/pmc_traffic.py)",
           "commit": os.environ.get("MV_COMMIT"),
           "lib_source_digest16": os.environ.get("MV_LIB_DIGEST"), "kernels": {}}
    for fam in FAMILIES:
        if fam in fetch and fam in write:
            nf, f = fetch[fam]
            nw, w = write[fam]
            res["kernels"][fam] = {"launches": nf, "fetch_kib_raw_avg": f, "write_kib_avg": w,
                                   "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
    with open(sys.argv[3], "w") as fh:
        json.dump(res, fh, indent=1)
    for k, v in res["kernels"].items():
        print(f"{k:10s} launches {v['launches']:5d}  fetch(raw) {v['fetch_kib_raw_avg'] / 1024:8.1f} MiB  write "
              f"{v['write_kib_avg'] / 1024:8.1f} MiB  -> {v['hbm_bytes_per_launch'] / 1e6:8.1f} MB per launch")


if __name__ == "__main__":
    main()
