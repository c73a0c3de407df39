#!/bin/bash
set -o pipefail
O=gpurun_out/r2e; mkdir -p $O
python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "fp32 or int8 or f32" 2>&1 | tail -30 | tee $O/newk.txt
python -m pytest tests/test_vit_parity.py -m gpu -x -q -k "micro_seg_fp16_32" 2>&1 | tail -60 > $O/segq.txt; tail -3 $O/segq.txt
python bench.py --workload infer-int8 --batch 1024 --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-1200 | tee $O/int8.txt
python bench.py --workload infer-int8 --batch 256 --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-200 | tee -a $O/int8.txt
