#!/bin/bash
set -o pipefail
O=gpurun_out/r2w; mkdir -p $O
python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "gemm_nt" 2>&1 | tail -4 | tee $O/tests.txt
python -m pytest tests/test_vit_parity.py -m gpu -x -q -k "bf16" 2>&1 | tail -3 | tee -a $O/tests.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330 | tee $O/bench.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-330 | tee -a $O/bench.txt
