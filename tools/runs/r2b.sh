#!/bin/bash
# round-2 GPU call B: full GPU test suite with measured-value report + per-shape in-step GEMM timings
set -o pipefail
O=gpurun_out/r2b; mkdir -p $O; rm -f $O/report.txt
MV_TEST_REPORT=$PWD/$O/report.txt python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/tests.txt
python bench.py --steps 12 --warmup 4 --per-shape --timer-every 2 --no-cpu-baseline > $O/bench_shape.json 2> $O/bench_shape.err
tail -c 3000 $O/bench_shape.json
