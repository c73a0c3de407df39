#!/bin/bash
# Diagnostic library for tools/diag/attn_timeline.py: attention.hip with -DMV_ATTN_TRACE=1, the other objects from the product build
set -e
cd "$(dirname "$0")/../.."
CS=myrtle-vision_amd/csrc
TL=$(python -c "import importlib.util,os;print(os.path.join(list(importlib.util.find_spec('torch').submodule_search_locations)[0],'lib'))")
mkdir -p tools/_ablate/st
for f in layernorm gemm_bf16 gemm_f32 elementwise seg_tail attention_f32 image_prep; do cp myrtle-vision_amd/lib/$f.o tools/_ablate/st/; done
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -DMV_ATTN_TRACE=1 -c $CS/attention.hip -o tools/_ablate/st/attention.o
g++ -shared -fPIC -o tools/_ablate/libattn_trace.so tools/_ablate/st/*.o -L$TL -l:libamdhip64.so -Wl,-rpath,$TL:/opt/rocm/lib
rm -rf tools/_ablate/st
