#!/bin/bash
# isa_cmp.sh old.o new.o : per-kernel comparison of gfx950 ISA modulo pc-relative constants
for f in "$1" "$2"; do b=$(echo $f | md5sum | cut -c1-8); /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=/tmp/kr/$b.fat $f </dev/null; /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/kr/$b.fat --output=/tmp/kr/$b.co --unbundle </dev/null; /opt/rocm/lib/llvm/bin/llvm-objdump -d /tmp/kr/$b.co </dev/null > /tmp/kr/$b.dis; done
A=$(echo $1 | md5sum | cut -c1-8); B=$(echo $2 | md5sum | cut -c1-8)
python3 - /tmp/kr/$A.dis /tmp/kr/$B.dis <<'PY'
import re,sys
def load(fn):
    d={}; cur=None
    for l in open(fn):
        m=re.match(r'^[0-9a-f]+ <(\S+)>:',l)
        if m: cur=m.group(1); d[cur]=[]; continue
        if cur is not None:
            t=l.split('//')[0].strip(); t=re.sub(r'<[^>]*>','',t)
            t=re.sub(r'(s_add_u32 s\d+, s\d+, )0x[0-9a-f]+',r'\1PCREL',t); t=re.sub(r'(s_addc_u32 s\d+, s\d+, )(0x[0-9a-f]+|-?\d+)',r'\1PCREL',t)
            if t: d[cur].append(t)
    return d
a=load(sys.argv[1]); b=load(sys.argv[2])
same=[k for k in a if k in b and a[k]==b[k]]; diff=[k for k in a if k in b and a[k]!=b[k]]
print("kernels same:",len(same),"different:",len(diff),"only old:",len([k for k in a if k not in b]),"only new:",len([k for k in b if k not in a]))
for k in diff[:8]:
    n=sum(1 for x,y in zip(a[k],b[k]) if x!=y); print("  DIFF",k[:70],len(a[k]),len(b[k]),n)
PY
