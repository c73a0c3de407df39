#!/bin/bash
# LDS counters of the attention kernels (bench_attn.py): bank conflicts vs LDS-active cycles
set -o pipefail
R=$PWD; O=$R/gpurun_out/r2aa; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d $O/p -o a -- python3 $R/tools/bench_attn.py > $O/out.txt 2> $O/err.txt
tail -1 $O/out.txt
ls $O/p | head
