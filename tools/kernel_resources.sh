#!/bin/bash
# registers / spills / scratch of the kernels in the built library (code-object metadata), optional name filter
# usage: tools/kernel_resources.sh [pattern]
so=$(realpath ${LQP_LIB:-lqp_py_amd/csrc/liblqp_amd.so})
tmp=$(mktemp -d)
cp $so $tmp/lib.so
(cd $tmp && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1)
# (the split build leaves one code object per translation unit)
for co in $(ls $tmp/lib.so.*gfx950* 2>/dev/null); do /opt/rocm/lib/llvm/bin/llvm-readelf --notes $co; done | python3 -c "
import sys,re
txt=sys.stdin.read()
pat=sys.argv[1] if len(sys.argv)>1 else ''
for blk in txt.split('- .agpr_count')[1:]:
    g=lambda k:(re.search(r'\.'+k+r':\s*(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    if pat in name:
        print(f\"{name[:80]:80s} vgpr {g('vgpr_count'):>4} sgpr {g('sgpr_count'):>4} vspill {g('vgpr_spill_count'):>4} sspill {g('sgpr_spill_count'):>4} scratch {g('private_segment_fixed_size'):>5}\")
" "$1"
rm -rf $tmp
