#!/bin/bash
# Build libmyrtle_vision_hip variants that differ only in layernorm.hip's experiment macros (LN_PREFETCH / LN_WAVES / LN_GRID),
# into myrtle-vision_amd/lib/ln_<name>.so (git-ignored; they travel to the GPU box).  Usage: ln_bwd_variants.sh name:flags ...
set -e
cd "$(dirname "$0")/../.."
python __graft_entry__.py > /dev/null
L=myrtle-vision_amd/lib
TL=$(python -c "import importlib.util,os;print(os.path.join(list(importlib.util.find_spec('torch').submodule_search_locations)[0],'lib'))")
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function $flags -I include -c myrtle-vision_amd/csrc/layernorm.hip -o /tmp/ln_$name.o
  objs=$(ls $L/*.o | grep -v layernorm.o)
  g++ -shared -fPIC -o $L/ln_$name.so $objs /tmp/ln_$name.o -L$TL -l:libamdhip64.so -Wl,-rpath,$TL:/opt/rocm/lib -Wl,--enable-new-dtags
  echo "built $L/ln_$name.so ($flags)"
done
