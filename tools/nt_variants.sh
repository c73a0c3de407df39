#!/bin/bash
# Experiment builds of libmyrtle_vision_hip with the MV_NT_* switches of gemm_bf16.hip (results stay CORRECT: only stores,
# tile order and start-up timing change).   build here: tools/nt_variants.sh "name:-Dflags" ... ; GPU box: MV_LIB_PATH=...
set -e
cd "$(dirname "$0")/.."
CS=myrtle-vision_amd/csrc
TL=$(python -c "import importlib.util,os;print(os.path.join(list(importlib.util.find_spec('torch').submodule_search_locations)[0],'lib'))")
mkdir -p tools/_ablate
build_one() {
  name=${1%%:*}; flags=${1#*:}
  d=tools/_ablate/o_$name; mkdir -p $d
  for f in layernorm attention gemm_f32 elementwise seg_tail image_prep; do cp myrtle-vision_amd/lib/$f.o $d/; done
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include $flags -c $CS/gemm_bf16.hip -o $d/gemm_bf16.o
  g++ -shared -fPIC -o tools/_ablate/libnt_$name.so $d/*.o -L$TL -l:libamdhip64.so -Wl,-rpath,$TL:/opt/rocm/lib
  rm -rf $d
  echo built tools/_ablate/libnt_$name.so
}
for v in "$@"; do build_one "$v" & done
wait
