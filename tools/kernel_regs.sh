#!/bin/bash
# registers / spills / LDS of the gfx950 kernels inside a hipcc object whose name matches a pattern:
#   tools/kernel_regs.sh advancedmh.jl_amd/csrc/mhx_api_f64.o k_ram_defer
set -e
export PATH=$PATH:/opt/rocm/lib/llvm/bin
o="$1"; pat="${2:-.}"
t=$(mktemp -d)
llvm-objcopy --dump-section .hip_fatbin=$t/fb "$o" $t/discard.o
clang-offload-bundler --type=o --unbundle --input=$t/fb --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$t/dev.co
llvm-readelf --notes $t/dev.co | python3 -c "
import sys, re
txt = sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    blk = '.agpr_count' + blk
    f = dict(re.findall(r'\.(\w+):\s+(\S+)', blk))
    if re.search(sys.argv[1], f.get('name', '')):
        print(f.get('name'), 'vgpr', f.get('vgpr_count'), 'agpr', f.get('agpr_count'), 'sgpr', f.get('sgpr_count'),
              'vgpr_spill', f.get('vgpr_spill_count'), 'sgpr_spill', f.get('sgpr_spill_count'), 'scratch', f.get('private_segment_fixed_size'),
              'lds', f.get('group_segment_fixed_size'))
" "$pat"
rm -rf $t
