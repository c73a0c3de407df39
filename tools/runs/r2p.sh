#!/bin/bash
set -o pipefail
O=gpurun_out/r2p; mkdir -p $O
python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "split3 or bf16x6 or bitwise or gemm_f32 or fp32_blocks" 2>&1 | tail -25 > $O/tests.txt; tail -4 $O/tests.txt
python tools/bench_f32.py 2>&1 | tee $O/bench_f32.txt &&
python bench.py --precision fp32 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400 | tee $O/fp32_step.txt &&
MV_F32_GEMM=mfma python bench.py --precision fp32 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200 | tee -a $O/fp32_step.txt &&
python -m pytest tests/test_vit_parity.py -m gpu -x -q -k "fp32" 2>&1 | tail -8 | tee $O/parity.txt
