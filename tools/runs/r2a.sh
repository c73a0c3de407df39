#!/bin/bash
# round-2 GPU call A: NT variant screen (sc1 stores, column bands, stagger, persistent) per shape and inside the step
set -o pipefail
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
echo "== default lib, variants 0 / 2569 ==" | tee $O/gemm.txt
NT_ONLY=1 ROTATE=4 NT_VARIANTS=0,2569 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee -a $O/gemm.txt
for v in sc1 band4 band6 band4sc1 stag4 stag4sc1; do
  echo "== lib $v ==" | tee -a $O/gemm.txt
  MV_LIB_PATH=$PWD/tools/_ablate/libnt_$v.so NT_ONLY=1 ROTATE=4 NT_VARIANTS=0 python tools/bench_gemm.py 2>&1 | grep "^NT" | tee -a $O/gemm.txt
done
echo "== correctness of variant libs (gemm op tests) ==" | tee $O/tests.txt
for v in band4sc1 stag4sc1; do
  MV_LIB_PATH=$PWD/tools/_ablate/libnt_$v.so python -m pytest tests/test_hip_ops.py tests/test_full_size_properties.py -m gpu -x -q -k "gemm or full" 2>&1 | tail -3 | tee -a $O/tests.txt
done
echo "== in-step A/B ==" | tee $O/step.txt
NT_VARIANTS=0,2569 ROUNDS=10 STEPS=10 python tools/ab_step.py 2>&1 | grep variant | tee -a $O/step.txt
for v in sc1 band4sc1 stag4sc1; do
  echo "lib $v:" | tee -a $O/step.txt
  MV_LIB_PATH=$PWD/tools/_ablate/libnt_$v.so NT_VARIANTS=0 ROUNDS=6 STEPS=10 python tools/ab_step.py 2>&1 | grep variant | tee -a $O/step.txt
done
echo "lib default again:" | tee -a $O/step.txt
NT_VARIANTS=0 ROUNDS=6 STEPS=10 python tools/ab_step.py 2>&1 | grep variant | tee -a $O/step.txt
