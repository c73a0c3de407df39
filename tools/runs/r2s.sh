#!/bin/bash
# round-2 evidence refresh: headline bench, per-kernel trace of the same command, seg256 / QAT / int8 / fp32 lines, DDP timeline
set -o pipefail
R=$PWD; O=$R/gpurun_out/r2s; mkdir -p $O
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench.json; cut -c1-400 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o step -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/prof_bench.txt 2>&1
cd $R
python tools/rocpd_kernel_stats.py $(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1) $O/kernel_stats.csv 8 | head -16 | cut -c1-150
rm -rf $O/prof
python bench.py --workload seg256 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/seg256.json; cut -c1-200 $O/seg256.json
python bench.py --workload seg --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/seg.json; cut -c1-200 $O/seg.json
python bench.py --q-format FP16_32 --batch 64 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/qat_fp16_32.json; cut -c1-200 $O/qat_fp16_32.json
python bench.py --q-format TF32 --batch 64 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/qat_tf32.json; cut -c1-200 $O/qat_tf32.json
python bench.py --q-format FP16_16 --batch 64 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/qat_fp16_16.json; cut -c1-200 $O/qat_fp16_16.json
python bench.py --workload infer-int8 --steps 8 --warmup 3 2>/dev/null | tail -1 > $O/int8.json; cut -c1-200 $O/int8.json
python tools/ddp_overlap_timeline.py > $O/ddp_timeline.txt 2>&1; tail -12 $O/ddp_timeline.txt
