#!/bin/bash
set -o pipefail
O=gpurun_out/r2k; mkdir -p $O; rm -f $O/report.txt
python tools/ddp_overlap_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/ddp_timeline.txt
MV_TEST_REPORT=$PWD/$O/report.txt python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/tests.txt; tail -4 $O/tests.txt
python bench.py --workload infer-int8 --steps 10 --warmup 3 2>&1 | tail -1 | cut -c1-700 | tee $O/int8.txt
python bench.py --workload seg --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-250 | tee $O/seg.txt
python bench.py --workload seg256 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-250 | tee -a $O/seg.txt
