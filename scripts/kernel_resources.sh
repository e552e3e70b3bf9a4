#!/bin/bash
# usage: scripts/kernel_resources.sh bilevel-gait-gen_amd/libsrbm_rti.so  -> per kernel of EVERY gfx950 code object of the library (one per .hip
# source: srbm_capi.hip): VGPRs, AGPRs, SGPRs, spills, scratch bytes per lane, static LDS
LIB=$1
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $LIB $T/fat.bin
python3 - $T <<'PY'
import re, subprocess, sys
T = sys.argv[1]
d = open(T + '/fat.bin', 'rb').read()
offs = [m.start() for m in re.finditer(b'__CLANG_OFFLOAD_BUNDLE__', d)]
for i, o in enumerate(offs):
    end = offs[i + 1] if i + 1 < len(offs) else len(d)
    open('%s/fat%d.bin' % (T, i), 'wb').write(d[o:end])
    subprocess.call(['/opt/rocm/lib/llvm/bin/clang-offload-bundler', '--unbundle', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                     '--input=%s/fat%d.bin' % (T, i), '--output=%s/k%d.co' % (T, i)])
    txt = subprocess.check_output(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', '%s/k%d.co' % (T, i)]).decode()
    print('# code object %d of %d' % (i + 1, len(offs)))
    for blk in txt.split('- .agpr_count:')[1:]:
        g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
        print('%-48s vgpr %s agpr %s sgpr %s vspill %s sspill %s scratch %s lds %s' % (g('name')[:48], g('vgpr_count'), blk.split()[0], g('sgpr_count'),
              g('vgpr_spill_count'), g('sgpr_spill_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
PY
rm -rf $T
