#!/bin/bash
# usage: scripts/kernel_resources.sh bilevel-gait-gen_amd/libsrbm_rti.so  -> per kernel of the gfx950 code object: VGPRs, AGPRs, SGPRs, spills, scratch bytes per lane, static LDS
LIB=$1
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$LIB --output=$T/k.co 2>/dev/null || { 
  # shared libs keep the fat binary in .hip_fatbin
  /opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $LIB $T/fat.bin
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co
}
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count:')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
    agpr=blk.split()[0]
    print('%-40s vgpr %s agpr %s sgpr %s vspill %s sspill %s scratch %s lds %s' % (g('name')[:40], g('vgpr_count'), agpr, g('sgpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
"
rm -rf $T
