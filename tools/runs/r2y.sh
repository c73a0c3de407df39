#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r2y; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o qat -- python3 $R/bench.py --q-format FP16_32 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.txt 2>&1
tail -1 $O/bench.txt | cut -c1-200
