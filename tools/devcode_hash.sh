#!/bin/bash
# md5 of the gfx950 device code (disassembly) inside a hipcc object: `tools/devcode_hash.sh advancedmh.jl_amd/csrc/mhx_api_f64.o`.
# A refactor of the kernel headers that must not change code generation (removing a knob whose default is kept) is checked by
# comparing this hash before and after -- no GPU needed.  -d keeps the disassembly next to the object (<obj>.dis).
set -e
export PATH=$PATH:/opt/rocm/lib/llvm/bin
keep=0; [ "$1" = "-d" ] && { keep=1; shift; }
for o in "$@"; do
  t=$(mktemp -d)
  llvm-objcopy --dump-section .hip_fatbin=$t/fb "$o" /dev/null 2>/dev/null || llvm-objcopy --dump-section .hip_fatbin=$t/fb "$o" $t/discard.o
  clang-offload-bundler --type=o --unbundle --input=$t/fb --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$t/dev.co
  llvm-objdump -d $t/dev.co | grep -v "file format" > $t/dis
  [ $keep = 1 ] && cp $t/dis "$o.dis"
  echo "$(md5sum < $t/dis | cut -d' ' -f1)  $o"
  rm -rf $t
done
