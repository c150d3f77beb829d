#!/bin/bash
# Register / scratch / LDS use of every kernel in a run-time compiled code object (.hsaco from the JIT cache):
#   bash tools/jit_regs.sh <file.hsaco>
f=$1
t=$(mktemp /tmp/jitregs.XXXXXX)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$f --output=$t
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $t | grep -E "\.name:|\.vgpr_count|\.sgpr_count|spill_count|private_segment_fixed|group_segment_fixed" | sed 's/^ *//' | paste - - - - - - - 
rm -f $t
