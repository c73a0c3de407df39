#!/bin/bash
set -o pipefail
O=gpurun_out/r2u; mkdir -p $O
python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "attention or fp32_blocks" 2>&1 | tail -6 > $O/tests.txt; tail -4 $O/tests.txt
python bench.py --precision fp32 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230 | tee $O/fp32_step.txt &&
python bench.py --precision fp32 --batch 256 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-230 | tee -a $O/fp32_step.txt &&
python -m pytest tests/test_vit_parity.py tests/test_train_gpu.py -m gpu -x -q -k "fp32 or FP16 or TF32 or quant" 2>&1 | tail -3 | tee $O/parity.txt
