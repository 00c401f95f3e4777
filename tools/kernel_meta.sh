#!/bin/bash
# tools/kernel_meta.sh OBJECT [NAME_SUBSTRING]: registers / scratch / LDS of the gfx950 kernels inside a HIP host object
# (objcopy .hip_fatbin -> clang-offload-bundler -> llvm-readelf --notes).
LLVM=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$1" $tmp/fat && $LLVM/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat --output=$tmp/co >/dev/null 2>&1
$LLVM/llvm-readelf --notes $tmp/co | python3 -c "
import sys,re
txt=sys.stdin.read()
pat=sys.argv[1] if len(sys.argv)>1 else ''
for blk in txt.split('  - .agpr_count')[1:]:
    g=lambda k:(re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    if pat in name:
        print('%-100s vgpr %s agpr %s sgpr %s scratch %s lds %s' % (name[:100], g('vgpr_count'), blk.split()[1] if blk.split() else '?', g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
" "$2"
rm -rf $tmp
