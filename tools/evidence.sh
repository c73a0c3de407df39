#!/bin/bash
# ONE script regenerates every number bench.py and DESIGN.md quote, from ONE source tree, and stamps the commit into each file
# (VERDICT round 2, item 8).  Runs on the GPU box:
#
#   COMMIT=$(git rev-parse --short HEAD) gpurun --timeout 1100 -- "COMMIT=$COMMIT bash tools/evidence.sh r03_b"
#   then:  cp gpurun_out/evidence_r03_b/r03_b_* profiles/
#
# Outputs (under gpurun_out/evidence_<tag>/, named <tag>_*):
#   _pmc_traffic.json   FETCH_SIZE / WRITE_SIZE passes (separate runs, gfx950 correction) -> HBM bytes per launch per family
#   _mfma_util.json     SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE pass -> MFMA-busy per family and for the attention+MLP block
#   _kernel_stats.csv   rocprofv3 --kernel-trace of the SAME bench command (per-kernel durations)
#   _bench.json         the headline line, run LAST so that it reads the PMC files written above (same tree, same commit)
#   _workload_lines.jsonl  the other bench.py workloads (seg, seg256, int8, batch 32, batch 64, fp32 at batch 64, bf16x3h, bf16x3, fp32, seg256 in both split modes)
#   _dist_rehearsal.jsonl  MV_FORCE_DIST=1 (one-rank RCCL group) with the fp32 and the bf16 gradient exchange
#   _stamp.json         commit, source digest of the loaded .so, date, rocm-smi clocks
# Counter passes use --kernel-trace + --pmc only (no sys-trace: refused on this pool).
set -o pipefail
TAG=${1:-r03_x}
R=$PWD; O=$R/gpurun_out/evidence_$TAG; mkdir -p $O
export MV_COMMIT=${COMMIT:-unknown}
DIGEST=$(cat $R/myrtle-vision_amd/lib/build.sha256 2>/dev/null | cut -c1-16)
export MV_LIB_DIGEST=$DIGEST      # stamped into the PMC summaries: bench.py quotes them only for the library they were measured on
python3 - > $O/${TAG}_stamp.json <<PY
import json, time
print(json.dumps({"commit": "$MV_COMMIT", "lib_source_digest16": "$DIGEST", "utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
                  "tag": "$TAG"}))
PY
BENCH="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o f -- python3 $BENCH > $O/pmc_f.txt 2>&1 || { echo "FETCH pass failed"; tail -5 $O/pmc_f.txt; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o w -- python3 $BENCH > $O/pmc_w.txt 2>&1 || { echo "WRITE pass failed"; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pu -o u -- python3 $BENCH > $O/pmc_u.txt 2>&1 || { echo "MFMA pass failed"; exit 1; }
rocprofv3 --kernel-trace --stats -d $O/pk -o k -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/trace_bench.txt 2>&1 || { echo "trace failed"; exit 1; }
cd $R
db() { ls $O/$1/*.db $O/$1/*/*.db 2>/dev/null | head -1; }
python3 tools/pmc_traffic.py $(db pf) $(db pw) $O/${TAG}_pmc_traffic.json
python3 tools/pmc_mfma_util.py $(db pu) $O/${TAG}_mfma_util.json
python3 tools/rocpd_kernel_stats.py $(db pk) $O/${TAG}_kernel_stats.csv 10 > $O/kernel_stats.txt 2>&1; sed -n 1,20p $O/kernel_stats.txt | cut -c1-150
rm -rf $O/pf $O/pw $O/pu $O/pk
# bench.py reads the NEWEST profiles/rNN_*; put this run's files there (on the box) so that the line below quotes this commit
cp $O/${TAG}_pmc_traffic.json $O/${TAG}_mfma_util.json $R/profiles/
python3 bench.py --steps 20 --warmup 5 2> $O/bench.err | tail -1 > $O/${TAG}_bench.json; cut -c1-300 $O/${TAG}_bench.json
: > $O/${TAG}_workload_lines.jsonl
for w in "--workload seg" "--workload seg256" "--workload infer-int8" "--batch 32 --steps 24" "--batch 64 --steps 16" "--precision fp32 --batch 64"; do
  python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline $w 2>> $O/bench.err | tail -1 >> $O/${TAG}_workload_lines.jsonl
done
# the reference-tolerance modes next to the headline
for w in "--precision bf16x3h" "--precision bf16x3" "--precision fp32" "--workload seg256 --precision bf16x3h" "--workload seg256 --precision bf16x3"; do
  python3 bench.py $w --steps 6 --warmup 2 --no-cpu-baseline 2>> $O/bench.err | tail -1 >> $O/${TAG}_workload_lines.jsonl
done
# launch-path record of the RCCL exchange with ONE rank (no multi-GPU hardware in this pool): both exchange dtypes
: > $O/${TAG}_dist_rehearsal.jsonl
for x in fp32 bf16; do
  MV_FORCE_DIST=1 MV_DDP_EXCHANGE=$x python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline 2>> $O/bench.err | tail -1 >> $O/${TAG}_dist_rehearsal.jsonl
done
cut -c1-170 $O/${TAG}_workload_lines.jsonl
python3 - <<PY
import json
for l in open("$O/${TAG}_dist_rehearsal.jsonl"):
    d = json.loads(l); print("dist rehearsal:", d["ms_per_step"], "ms", d.get("dist"))
PY
