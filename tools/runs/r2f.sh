#!/bin/bash
# profile the int8 inference workload and the fp32 training step per kernel
set -o pipefail
O=gpurun_out/r2f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_ops.py tests/test_vit_parity.py -m gpu -x -q -k "fp32 or int8 or f32 or micro_seg_fp16_32" 2>&1 | tail -4 | tee $O/tests.txt
rocprofv3 --kernel-trace --stats -d $O/prof_int8 -o int8 -- python3 bench.py --workload infer-int8 --batch 1024 --steps 4 --warmup 2 > $O/int8.json 2> $O/int8.err
python tools/rocpd_kernel_stats.py $(ls $O/prof_int8/*/*.db $O/prof_int8/*.db 2>/dev/null | head -1) $O/int8_kernel_stats.csv 4 | tee $O/int8_stats.txt
rocprofv3 --kernel-trace --stats -d $O/prof_fp32 -o fp32 -- python3 bench.py --precision fp32 --batch 64 --steps 4 --warmup 2 --no-cpu-baseline > $O/fp32.json 2> $O/fp32.err
python tools/rocpd_kernel_stats.py $(ls $O/prof_fp32/*/*.db $O/prof_fp32/*.db 2>/dev/null | head -1) $O/fp32_kernel_stats.csv 4 | tee $O/fp32_stats.txt
rm -rf $O/prof_int8 $O/prof_fp32
