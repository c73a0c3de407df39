#!/bin/bash
# Diagnostic: build libmyrtle_vision_hip variants with one phase of attn_bwd4_kernel removed (results are WRONG by
# construction) and time each with tools/bench_attn.py.   usage (GPU box): tools/ablate_attn.sh run
set -e
cd "$(dirname "$0")/.."
CS=myrtle-vision_amd/csrc
if [ "$1" != "run" ]; then
  for m in ${MASKS:-1 2 4 8 16 31}; do
    mkdir -p tools/_ablate/o$m
    for f in layernorm gemm_bf16 gemm_f32 elementwise seg_tail attention_f32 image_prep; do cp myrtle-vision_amd/lib/$f.o tools/_ablate/o$m/; done
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -DMV_ATTN_ABLATE=$m -c $CS/attention.hip -o tools/_ablate/o$m/attention.o
    TL=$(python -c "import importlib.util,os;print(os.path.join(list(importlib.util.find_spec('torch').submodule_search_locations)[0],'lib'))")
    g++ -shared -fPIC -o tools/_ablate/libattn_ablate$m.so tools/_ablate/o$m/*.o -L$TL -l:libamdhip64.so -Wl,-rpath,$TL:/opt/rocm/lib
    rm -rf tools/_ablate/o$m
  done
else
  python tools/bench_attn.py
  for m in ${MASKS:-1 2 4 8 16 31}; do echo "ablate mask $m:"; MV_LIB_PATH=$PWD/tools/_ablate/libattn_ablate$m.so python tools/bench_attn.py; done
fi
