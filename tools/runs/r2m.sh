#!/bin/bash
set -o pipefail
O=gpurun_out/r2m; mkdir -p $O; rm -f $O/report.txt
MV_TEST_REPORT=$PWD/$O/report.txt python -m pytest tests/test_vit_parity.py -m gpu -x -q -k "fake_quant" 2>&1 | tail -25 > $O/fq.txt; tail -4 $O/fq.txt; grep "FP16\|TF32" $O/report.txt
python -m pytest tests/test_hip_ops.py tests/test_train_gpu.py -m gpu -x -q -k "f16 or fp16 or quantized_evaluation" 2>&1 | tail -25 > $O/tests.txt; tail -4 $O/tests.txt
python bench.py --q-format FP16_32 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900 | tee $O/fp16_32.txt
