#!/bin/bash
set -o pipefail
O=gpurun_out/r2g; mkdir -p $O
python -m pytest tests/test_hip_ops.py tests/test_vit_parity.py -m gpu -x -q -k "fp32 or int8 or f32 or micro_seg_fp16_32" 2>&1 | tail -40 > $O/tests.txt; tail -4 $O/tests.txt
python tools/bench_f32.py 2>&1 | tee $O/f32bench.txt
python bench.py --precision fp32 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300 | tee $O/fp32step.txt
python bench.py --workload infer-int8 --batch 1024 --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-300 | tee $O/int8.txt
