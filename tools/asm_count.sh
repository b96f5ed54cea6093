#!/bin/bash
# Static instruction counts of selected kernels (no GPU needed): tools/asm_count.sh <pattern>
# prints: total instructions, instructions before the first s_barrier, scalar loads, SGPR-spill lane moves
mkdir -p /tmp/dis && cd /tmp/dis
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -S --cuda-device-only -I/root/repo/include /root/repo/causal-gen_amd/csrc/conv.hip -o conv.s 2>/dev/null
python3 - "$1" <<'PY'
import re, sys, subprocess
pat = sys.argv[1]
lines = open('/tmp/dis/conv.s').read().split('\n')
i = 0
while i < len(lines):
    m = re.match(r'^(_Z\w+):', lines[i])
    if m and pat in m.group(1):
        j = i + 1
        while not lines[j].startswith('.Lfunc_end'): j += 1
        ins = [l.strip() for l in lines[i + 1:j] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
        fb = next((n for n, x in enumerate(ins) if x.startswith('s_barrier')), len(ins))
        dem = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        pre = ins[:fb]
        print('%-48s total %5d  pre-barrier %5d  s_load %3d  lane-spill %3d  v_mul_lo/hi %3d  branches %3d' % (
            re.sub(r'\(.*', '', dem).replace('void cgen::', '')[:48], len(ins), fb, sum(x.startswith('s_load') for x in pre),
            sum(x.startswith(('v_writelane', 'v_readlane')) for x in pre), sum(x.startswith(('v_mul_lo', 'v_mul_hi')) for x in pre),
            sum(x.startswith('s_cbranch') for x in pre)))
        i = j
    i += 1
PY
