#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage for the csrc/*.hip kernels."""
import re, subprocess, sys, os
src = sys.argv[1:]
for f in src:
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", f, "-o", "/dev/null",
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: (?:Function Name: (\S+)|\s*([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+))", line)
        if not m: continue
        if m.group(1):
            if cur: print(cur)
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"kernel": name[:90]}
        else:
            k = m.group(2).strip()
            if k in ("VGPRs", "AGPRs", "VGPRs Spill", "ScratchSize", "Occupancy", "LDS Size", "SGPRs"):
                cur[k] = int(m.group(3))
    if cur: print(cur)
