#!/bin/bash
# The phase-ablation builds (MV_ABLATE, MV_ATTN_ABLATE, MV_AF_ABLATE), the per-workgroup timeline stamps (MV_ABLATE=32,
# MV_ATTN_TRACE) and the retired A/B arms of rounds 1-3 (NT 2565 / 2567 / 2569, TN 2565) are NOT part of the product sources any
# more (round 4).  They live in the history: commit a3a41b8 is the last tree that carries them, together with the scripts that
# build and run them (tools/ablate_gemm8.sh, tools/ablate_attn.sh, tools/ablate_attn_f32.sh, tools/diag/build_attn_trace.sh,
# tools/diag/p8_timeline.py, tools/diag/attn_timeline.py).  This script checks that tree out next to the repository and runs one of
# its scripts there; the C ABI of that revision differs (65 entry points), so its own Python package is used with it.
#
#   tools/diag/instrumented.sh tools/ablate_gemm8.sh            # build the ablated libraries (here, no GPU needed)
#   MODES="64 128 256" tools/diag/instrumented.sh tools/ablate_gemm8.sh
set -e
REV=${REV:-a3a41b8}
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
WT="$ROOT/tools/_ablate/worktree_$REV"
if [ ! -d "$WT" ]; then
  mkdir -p "$ROOT/tools/_ablate"
  git -C "$ROOT" worktree add --detach "$WT" "$REV"
fi
cd "$WT"
python __graft_entry__.py > /dev/null          # its product library first: the scripts reuse the unchanged objects
exec "$@"
