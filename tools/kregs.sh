#!/bin/bash
# VGPR / SGPR / spill / LDS figures of the kernels in an object file (the code object's metadata notes): tools/kregs.sh <obj.o> [name filter]
O=$1; F=${2:-.}
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $O </dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co --unbundle </dev/null 2>/dev/null
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co 2>/dev/null | python3 -c "
import re,sys,subprocess
t=sys.stdin.read()
for m in re.finditer(r'\.name:\s+(\S+)(.*?)\.wavefront_size', t, re.S):
    n=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip()
    b=m.group(2)
    if not re.search(r'''$F''', n): continue
    g=lambda k: (re.search(k+r':\s+(\d+)', b) or [0,'?'])[1]
    print(n.split('(')[0][-80:], 'vgpr',g(r'\.vgpr_count'),'spill',g(r'\.vgpr_spill_count'),'sgpr',g(r'\.sgpr_count'),'scratch',g(r'\.private_segment_fixed_size'))
"
rm -rf $T
