#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out/r2q; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o fp32 -- python3 $R/bench.py --precision fp32 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.txt 2>&1
tail -1 $O/bench.txt | cut -c1-200
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/fp32_kernel_stats.csv
head -30 $O/fp32_kernel_stats.csv | cut -c1-220
