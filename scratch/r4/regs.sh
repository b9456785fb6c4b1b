#!/bin/bash
# VGPR / scratch use of the kernels in an object file: bash scratch/r4/regs.sh conv16 [name filter]
O=/root/repo/spr_pick_amd/csrc/_build/$1.o
B=/opt/rocm/lib/llvm/bin
$B/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$O --output=/tmp/$1.co 2>/dev/null
$B/llvm-readelf --notes /tmp/$1.co | python3 -c "
import sys,re
txt=sys.stdin.read()
for m in re.finditer(r'\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)', txt, re.S):
    n=m.group(1)
    if '$2' in n: print('%-110s scratch %5s sgpr %4s vgpr %4s'%(n[:110], m.group(2), m.group(3), m.group(4)))
"
