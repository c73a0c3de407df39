#!/bin/bash
# Diagnostic: libmyrtle_vision_hip variants with one phase of attn_fwd_f32_kernel removed (results WRONG by construction),
# timed with tools/bench_attn_f32.py.   usage: tools/ablate_attn_f32.sh (build, here)  |  tools/ablate_attn_f32.sh run (GPU box)
set -e
cd "$(dirname "$0")/.."
CS=myrtle-vision_amd/csrc
if [ "$1" != "run" ]; then
  for m in ${MASKS:-1 2 4 5 8 16 31}; do
    mkdir -p tools/_ablate/o$m
    for f in layernorm gemm_bf16 gemm_f32 elementwise seg_tail attention image_prep; do cp myrtle-vision_amd/lib/$f.o tools/_ablate/o$m/; done
    hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -DMV_AF_ABLATE=$m -c $CS/attention_f32.hip -o tools/_ablate/o$m/attention_f32.o
    TL=$(python -c "import importlib.util,os;print(os.path.join(list(importlib.util.find_spec('torch').submodule_search_locations)[0],'lib'))")
    g++ -shared -fPIC -o tools/_ablate/libaf_ablate$m.so tools/_ablate/o$m/*.o -L$TL -l:libamdhip64.so -Wl,-rpath,$TL:/opt/rocm/lib
    rm -rf tools/_ablate/o$m
  done
else
  python tools/bench_attn_f32.py
  for m in ${MASKS:-1 2 4 5 8 16 31}; do echo "ablate mask $m:"; MV_LIB_PATH=$PWD/tools/_ablate/libaf_ablate$m.so python tools/bench_attn_f32.py; done
fi
