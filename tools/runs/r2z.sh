#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r2z; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o b32 -- python3 $R/bench.py --batch ${B:-32} --steps 10 --warmup 4 --no-cpu-baseline > $O/bench.txt 2>&1
tail -1 $O/bench.txt | cut -c1-200
