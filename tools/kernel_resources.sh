#!/bin/bash
# VGPR / scratch use of the kernels in a hipcc object file:  tools/kernel_resources.sh sweep_f64.o [name-filter]
B=/opt/rocm/lib/llvm/bin
t=$(mktemp -d)
$B/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $t/fat
$B/clang-offload-bundler --unbundle --type=o --input=$t/fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$t/co
$B/llvm-readelf --notes $t/co | python3 -c "
import sys,re
txt=sys.stdin.read()
flt=sys.argv[1] if len(sys.argv)>1 else ''
for blk in txt.split('- .agpr_count')[1:]:
    g=lambda k: re.search(r'\.'+k+r':\s+(\S+)', blk).group(1)
    n=g('name')
    if flt in n: print('%-90s vgpr %3s sgpr %3s scratch %5s lds %6s' % (n[-90:], g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
" "${2:-}"
rm -rf $t
