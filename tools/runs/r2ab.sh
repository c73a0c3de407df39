#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r2ab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 0 1 16; do
  if [ $m = 0 ]; then unset MV_LIB_PATH; else export MV_LIB_PATH=$R/tools/_ablate/libattn_ablate$m.so; fi
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d $O/p$m -o a -- python3 $R/tools/bench_attn.py > $O/out$m.txt 2> $O/err$m.txt
  ATTN_BWD=2 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS -d $O/q$m -o a -- python3 $R/tools/bench_attn.py > $O/outq$m.txt 2> $O/errq$m.txt
done
echo done
