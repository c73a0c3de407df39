#!/bin/bash
set -o pipefail
O=gpurun_out/r2h; mkdir -p $O; rm -f $O/report.txt
python tools/bench_f32.py 2>&1 | tee $O/f32bench.txt
python bench.py --precision fp32 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300 | tee $O/fp32step.txt
MV_TEST_REPORT=$PWD/$O/report.txt python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > $O/tests.txt; tail -5 $O/tests.txt
