#!/bin/bash
# tools/asm_variant.sh NAME DW_EXP PATCH.py -- development only: compile din_wave.hip (-DDW_EXP) to gfx950 assembly, let PATCH.py (reads
# the .s path in argv[1], edits in place) change it, assemble and link the result with the other objects into libdir_hip_<NAME>.so.
set -e
NAME=$1; EXP=$2; PATCH=$3
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/details-in-recommendation_amd; T=/tmp/asmv_$NAME; mkdir -p $T
LL=/opt/rocm/lib/llvm/bin
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -x hip -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None -DDW_EXP=$EXP"
/opt/rocm/bin/hipcc $FL -S --cuda-device-only $P/csrc/din_wave.hip -o $T/dev.s 2>/dev/null
[ -n "$PATCH" ] && python3 $PATCH $T/dev.s
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/dev.s -o $T/dev.o
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/dev.out $T/dev.o
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/dev.out -output=$T/dev.hipfb
/opt/rocm/bin/hipcc $FL --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -c $P/csrc/din_wave.hip -o $T/host.o 2>/dev/null
objs=$(ls $P/csrc/_build/*.o | grep -v "din_wave")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libdir_hip_$NAME.so $objs $T/host.o
echo "built libdir_hip_$NAME.so"
