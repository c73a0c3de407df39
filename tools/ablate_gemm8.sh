#!/bin/bash
# Diagnostic: libmyrtle_vision_hip variants with parts of gemm_nt_8phase_kernel removed (results are WRONG by
# construction): 8 = no epilogue, 16 = no main loop; 32 = whole kernel with per-workgroup
# time stamps (tools/diag/p8_timeline.py; MODES=32 tools/ablate_gemm8.sh); main loop only, bits may be combined: 64 = no fragment reads, 128 = no DMA, 256 = no MFMAs (320 = the DMA stream and its barriers alone).   build here: tools/ablate_gemm8.sh ; GPU box: tools/ablate_gemm8.sh run
set -e
cd "$(dirname "$0")/.."
CS=myrtle-vision_amd/csrc
if [ "$1" != "run" ]; then
  for m in ${MODES:-8 16}; do
    mkdir -p tools/_ablate/o$m
    for f in layernorm attention attention_f32 gemm_f32 elementwise seg_tail image_prep; do cp myrtle-vision_amd/lib/$f.o tools/_ablate/o$m/; done
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -DMV_ABLATE=$m -c $CS/gemm_bf16.hip -o tools/_ablate/o$m/gemm_bf16.o
    TL=$(python -c "import importlib.util,os;print(os.path.join(list(importlib.util.find_spec('torch').submodule_search_locations)[0],'lib'))")
    g++ -shared -fPIC -o tools/_ablate/libgemm_ablate$m.so tools/_ablate/o$m/*.o -L$TL -l:libamdhip64.so -Wl,-rpath,$TL:/opt/rocm/lib
    rm -rf tools/_ablate/o$m
  done
else
  NT_ONLY=1 NT_VARIANTS=0 python tools/bench_gemm.py 2>&1 | grep "^NT"
  for m in ${MODES:-8 16}; do echo "ablate $m:"; MV_LIB_PATH=$PWD/tools/_ablate/libgemm_ablate$m.so NT_ONLY=1 NT_VARIANTS=0 python tools/bench_gemm.py 2>&1 | grep "^NT"; done
fi
