#!/bin/bash
set -o pipefail
O=gpurun_out/r2d; mkdir -p $O
python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "f32 or int8 or materialised" 2>&1 | tail -30 | tee $O/newk.txt
python -m pytest tests/test_vit_parity.py -m gpu -x -q -k "fp32_matches or taps or fake_quant" 2>&1 | tail -5 | tee $O/fp32.txt
python tools/bench_f32.py 2>&1 | tee $O/f32bench.txt
python bench.py --workload infer-int8 --batch 1024 --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-1200 | tee $O/int8.txt
python bench.py --precision fp32 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300 | tee $O/fp32step.txt
