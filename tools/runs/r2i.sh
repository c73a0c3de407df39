#!/bin/bash
set -o pipefail
O=gpurun_out/r2i; mkdir -p $O
echo "== default ==" | tee $O/gemm.txt
NT_ONLY=1 ROTATE=4 NT_VARIANTS=0 python tools/bench_gemm.py 2>&1 | grep "^NT" | tee -a $O/gemm.txt
echo "== ldsst ==" | tee -a $O/gemm.txt
MV_LIB_PATH=$PWD/tools/_ablate/libnt_ldsst.so NT_ONLY=1 ROTATE=4 NT_VARIANTS=0 python tools/bench_gemm.py 2>&1 | grep "^NT" | tee -a $O/gemm.txt
MV_LIB_PATH=$PWD/tools/_ablate/libnt_ldsst.so python -m pytest tests/test_hip_ops.py tests/test_full_size_properties.py -m gpu -x -q -k "gemm or full" 2>&1 | tail -3 | tee $O/tests.txt
echo "default:" | tee $O/step.txt
NT_VARIANTS=0 ROUNDS=6 STEPS=10 python tools/ab_step.py 2>&1 | grep variant | tee -a $O/step.txt
echo "ldsst:" | tee -a $O/step.txt
MV_LIB_PATH=$PWD/tools/_ablate/libnt_ldsst.so NT_VARIANTS=0 ROUNDS=6 STEPS=10 python tools/ab_step.py 2>&1 | grep variant | tee -a $O/step.txt
