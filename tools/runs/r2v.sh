#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r2v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o int8 -- python3 $R/bench.py --workload infer-int8 --steps 5 --warmup 2 > $O/bench.txt 2>&1
tail -1 $O/bench.txt | cut -c1-200
