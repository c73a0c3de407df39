#!/bin/bash
set -o pipefail
O=gpurun_out/r2l; mkdir -p $O
python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "f16 or fp16" 2>&1 | tail -25 > $O/tests.txt; tail -4 $O/tests.txt
python -m pytest tests/test_vit_parity.py -m gpu -x -q -k "fake_quant" 2>&1 | tail -4 | tee -a $O/tests.txt
python bench.py --q-format FP16_32 --batch 64 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900 | tee $O/fp16_32.txt
python tools/ddp_overlap_timeline.py 2>&1 | grep -v "amdgpu.ids\|c10d\|version\|Hostname\|Librccl" | tee $O/ddp_timeline.txt
