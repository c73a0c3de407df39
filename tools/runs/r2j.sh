#!/bin/bash
# round-2 profiling passes: kernel stats, HBM traffic (2 passes), MFMA utilisation, per-shape fetch
set -o pipefail
O=gpurun_out/r2j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/ks -o ks -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_prof.json 2> $O/ks.err
python tools/rocpd_kernel_stats.py $(ls $O/ks/*/*.db $O/ks/*.db 2>/dev/null | head -1) $O/kernel_stats.csv 13 > $O/kernel_stats.txt; head -16 $O/kernel_stats.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o f -- $B > /dev/null 2> $O/pf.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o w -- $B > /dev/null 2> $O/pw.err
python tools/pmc_traffic.py $(ls $O/pf/*/*.db $O/pf/*.db 2>/dev/null | head -1) $(ls $O/pw/*/*.db $O/pw/*.db 2>/dev/null | head -1) $O/pmc_traffic.json | tee $O/traffic.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pu -o u -- $B > /dev/null 2> $O/pu.err
python tools/pmc_mfma_util.py $(ls $O/pu/*/*.db $O/pu/*.db 2>/dev/null | head -1) $O/mfma_util.json | tee $O/util.txt
NT_ONLY=1 ROTATE=4 NT_VARIANTS=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/ps -o s -- python3 tools/bench_gemm.py > $O/shape_bench.txt 2> $O/ps.err
python tools/pmc_per_shape.py $(ls $O/ps/*/*.db $O/ps/*.db 2>/dev/null | head -1) FETCH_SIZE gemm_nt | tee $O/pmc_nt_per_shape.txt
rm -rf $O/ks $O/pf $O/pw $O/pu $O/ps
python bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; tail -c 600 $O/bench_final.json
