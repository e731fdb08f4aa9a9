#!/bin/bash
# Register / scratch / LDS use of the kernels of one translation unit (from the code-object metadata of the built object).
# usage: scripts/kernel_regs.sh tq_cosmos [name-regex]
R=$(cd $(dirname $0)/.. && pwd)
O=$R/tapqir_amd/build/$1.o
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin $O $T/fatbin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fatbin --output=$T/k.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co | python3 -c "
import sys,re,subprocess
txt=sys.stdin.read()
pat=sys.argv[1] if len(sys.argv)>1 else ''
for blk in txt.split('- .agpr_count')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s*(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    dn=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()
    dn=re.sub(r'\(.*','',dn).replace('void ','')
    if pat and not re.search(pat,dn): continue
    print(f\"{dn[:64]:64s} vgpr={g('vgpr_count'):>4s} sgpr={g('sgpr_count'):>4s} spill={g('vgpr_spill_count'):>4s} scratch={g('private_segment_fixed_size'):>5s} lds={g('group_segment_fixed_size'):>6s}\")
" "$2"
cp $T/k.co /tmp/$1.co
rm -rf $T
