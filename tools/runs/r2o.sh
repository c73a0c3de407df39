#!/bin/bash
set -o pipefail
O=gpurun_out/r2o; mkdir -p $O
python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "int8 or fused_quantiser" 2>&1 | tail -25 > $O/tests.txt; tail -4 $O/tests.txt
python bench.py --workload infer-int8 --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-800 | tee $O/int8.txt
python bench.py --workload infer-int8 --batch 256 --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-200 | tee -a $O/int8.txt
