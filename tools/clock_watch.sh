#!/bin/bash
# Sample the GPU's shader clock, power and temperature while bench.py runs (evidence for the DVFS notes in DESIGN.md).
# usage (on the GPU box): bash tools/clock_watch.sh gpurun_out/clock_default.log [bench args / env before]
#                         CMD="python3 tools/clock_kernel_loop.py nt_qkv" bash tools/clock_watch.sh gpurun_out/clock_nt.log
out=$1; shift
mkdir -p "$(dirname "$out")"
${CMD:-python3 bench.py --steps ${STEPS:-200} --warmup 5 --no-cpu-baseline} "$@" > "$out.bench" 2>&1 &
pid=$!
: > "$out"
while kill -0 $pid 2>/dev/null; do
  { date +%s.%N; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|edge)" ; } >> "$out"
  sleep 0.4
done
wait $pid
tail -1 "$out.bench"
