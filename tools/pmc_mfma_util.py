#!/usr/bin/env python3
"""MFMA utilisation per kernel family from ONE rocprofv3 PMC pass (rocpd database):

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d out -o u -- \
        python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/pmc_mfma_util.py out/u_results.db profiles/rNN_mfma_util.json

utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 reports
the sum over the 8 XCDs; MI355X_MICROARCH.md, DVFS give-back).  It is the fraction of SIMD-cycles in which the matrix pipe
was executing -- a per-CLOCK figure: multiply by the clock the kernel actually held to compare with the 2.5 PFLOP/s peak
(which assumes 2.4 GHz).  bench.py reads ``kernels.<family>.mfma_util`` and ``block.mfma_util`` from the resulting file.
"""
import json
import os
import sqlite3
import sys

FAMILIES = {"gemm_nt": "gemm_nt_", "gemm_tn": "gemm_tn_", "attention": "attn_", "layernorm": "ln_"}
SIMDS = 1024


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select dispatch_id, name, counter_name, counter_value from pmc_events").fetchall()
    per = {}
    for did, name, cname, val in rows:
        d = per.setdefault(did, {"name": name})
        d[cname] = d.get(cname, 0.0) + val
    fam = {k: {"launches": 0, "mfma_busy": 0.0, "cycles": 0.0, "sq_busy": 0.0} for k in FAMILIES}
    tot = {"mfma_busy": 0.0, "cycles": 0.0}
    for d in per.values():
        cyc = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        tot["mfma_busy"] += busy
        tot["cycles"] += cyc
        for k, key in FAMILIES.items():
            if key in d["name"]:
                f = fam[k]
                f["launches"] += 1
                f["mfma_busy"] += busy
                f["cycles"] += cyc
                f["sq_busy"] += d.get("SQ_BUSY_CYCLES", 0.0)
    out = {"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over bench.py "
                     "--steps 2 --warmup 1 --no-cpu-baseline (tools/pmc_mfma_util.py); util = MFMA busy cycles / (GRBM_GUI_ACTIVE/8 "
                     "x 1024 SIMDs)", "commit": os.environ.get("MV_COMMIT"),
           "lib_source_digest16": os.environ.get("MV_LIB_DIGEST"), "kernels": {}}
    for k, f in fam.items():
        if f["launches"]:
            out["kernels"][k] = {"launches": f["launches"], "cycles_per_launch": f["cycles"] / f["launches"],
                                 "mfma_util": f["mfma_busy"] / max(f["cycles"] * SIMDS, 1.0)}
    blk = [fam[k] for k in ("gemm_nt", "gemm_tn", "attention", "layernorm")]
    out["block"] = {"what": "attention + MLP blocks = every NT / TN GEMM, attention and LayerNorm launch of the step",
                    "mfma_util": sum(f["mfma_busy"] for f in blk) / max(sum(f["cycles"] for f in blk) * SIMDS, 1.0)}
    out["all_kernels"] = {"mfma_util": tot["mfma_busy"] / max(tot["cycles"] * SIMDS, 1.0)}
    with open(sys.argv[2], "w") as fh:
        json.dump(out, fh, indent=1)
    for k, v in out["kernels"].items():
        print(f"{k:10s} launches {v['launches']:5d}  {v['cycles_per_launch']:10.0f} cycles/launch  MFMA util {v['mfma_util']:.3f}")
    print(f"block (GEMMs + attention + LayerNorm): MFMA util {out['block']['mfma_util']:.3f};  all kernels {out['all_kernels']['mfma_util']:.3f}")


if __name__ == "__main__":
    main()
