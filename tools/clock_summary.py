"""Summarise tools/clock_watch.sh logs: mean shader clock / package power over the samples taken while the step was
running (power > 900 W), next to the bench line of the same run.   usage: python tools/clock_summary.py gpurun_out/clock_*.log"""
import json
import re
import sys

print("| run | busy samples | sclk MHz (mean, min-max) | package W (mean) | img/s | NT TFLOP/s | TN TFLOP/s |")
print("|---|---|---|---|---|---|---|")
for path in sys.argv[1:]:
    t = open(path).read()
    s = [int(x) for x in re.findall(r"sclk clock level: \w+: \((\d+)Mhz", t)]
    p = [float(x) for x in re.findall(r"Package Power \(W\): ([\d.]+)", t)]
    busy = [(a, b) for a, b in zip(s, p) if b > 900]
    b = json.loads(open(path + ".bench").read().strip().splitlines()[-1])
    tn = [v for k, v in b["kernels"].items() if k.startswith("gemm_tn")][0]["tflops"]
    cl = [a for a, _ in busy]
    print(f"| {path.split('/')[-1]} | {len(busy)} | {sum(cl) / len(cl):.0f} ({min(cl)}-{max(cl)}) | "
          f"{sum(w for _, w in busy) / len(busy):.0f} | {b['value']:.0f} | {b['roofline']['achieved']:.0f} | {tn:.0f} |")
