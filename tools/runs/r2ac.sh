#!/bin/bash
# LDS bank-conflict counters for every kernel of the training step
set -o pipefail
R=$PWD; O=$R/gpurun_out/r2ac; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT -d $O/p -o a -- python3 $R/bench.py ${BENCH_ARGS:---steps 2 --warmup 1} --no-cpu-baseline > $O/out.txt 2> $O/err.txt
echo done
