#!/bin/bash
# the non-headline bench.py workloads, full JSON lines (profiles/r02_b_workload_lines.jsonl) + fp32 mode
set -o pipefail
O=gpurun_out/r2s; mkdir -p $O
python bench.py --workload seg256 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/seg256.json
python bench.py --workload seg --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/seg.json
python bench.py --q-format FP16_32 --batch 64 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/qat_fp16_32.json
python bench.py --q-format TF32 --batch 64 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/qat_tf32.json
python bench.py --q-format FP16_16 --batch 64 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/qat_fp16_16.json
python bench.py --workload infer-int8 --steps 8 --warmup 3 2>/dev/null | tail -1 > $O/int8.json
python bench.py --precision fp32 --batch 64 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/fp32_b64.json
python bench.py --precision fp32 --batch 256 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/fp32_b256.json
for f in seg256 seg qat_fp16_32 qat_tf32 qat_fp16_16 int8 fp32_b64 fp32_b256; do cut -c1-160 $O/$f.json; done
