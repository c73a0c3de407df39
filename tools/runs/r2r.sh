#!/bin/bash
set -o pipefail
O=gpurun_out/r2r; mkdir -p $O
python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "split3 or bf16x6 or fp32_blocks" 2>&1 | tail -5 > $O/tests.txt; tail -3 $O/tests.txt
python bench.py --precision fp32 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230 | tee $O/fp32_step.txt &&
python bench.py --precision fp32 --batch 256 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230 | tee -a $O/fp32_step.txt &&
MV_F32_GEMM=mfma python bench.py --precision fp32 --batch 256 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230 | tee -a $O/fp32_step.txt &&
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/full_gpu_tests.txt
