#!/bin/bash
set -o pipefail
O=gpurun_out/r2n; mkdir -p $O
python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "variant or epilogue" 2>&1 | tail -5 | tee $O/tests.txt
NT_ONLY=1 ROTATE=4 NT_VARIANTS=0,2562 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | tee $O/gemm.txt
NT_VARIANTS=0,2562 ROUNDS=6 STEPS=10 python tools/ab_step.py 2>&1 | grep variant | tee $O/step.txt
