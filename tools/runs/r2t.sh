#!/bin/bash
# end-of-round PMC refresh: HBM traffic (2 passes) and MFMA utilisation of the final kernels
set -o pipefail
O=gpurun_out/r2t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o f -- $B > /dev/null 2> $O/pf.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o w -- $B > /dev/null 2> $O/pw.err
python tools/pmc_traffic.py $(ls $O/pf/*/*.db $O/pf/*.db 2>/dev/null | head -1) $(ls $O/pw/*/*.db $O/pw/*.db 2>/dev/null | head -1) $O/pmc_traffic.json | tee $O/traffic.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pu -o u -- $B > /dev/null 2> $O/pu.err
python tools/pmc_mfma_util.py $(ls $O/pu/*/*.db $O/pu/*.db 2>/dev/null | head -1) $O/mfma_util.json | tee $O/util.txt
rm -rf $O/pf $O/pw $O/pu
