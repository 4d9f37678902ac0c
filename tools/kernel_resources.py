"""Compile one csrc/*.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per kernel:
VGPRs, AGPRs, spills, scratch, LDS, occupancy.  Usage: python tools/kernel_resources.py sepconv.hip [filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'empanada-napari_amd', 'csrc', sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ''
extra = ['-ffp-contract=off', '-fhip-fp32-correctly-rounded-divide-sqrt'] if sys.argv[1] == 'postprocess.hip' else []
out = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', '/tmp/_kr.o',
                      '-Rpass-analysis=kernel-resource-usage'] + extra, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'remark: [^ ]+ (Function Name|Name): (\S+)', line) or re.search(r'(Function Name|Name): (\S+)', line)
    if m:
        cur = subprocess.run(['c++filt', m.group(2)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r'\s+(VGPRs|AGPRs|VGPRs Spill|SGPRs|SGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)', line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        name = re.sub(r'\(anonymous namespace\)::', '', k)
        name = re.sub(r'\(.*', '', name)[-70:]
        print('%-70s vgpr %3d agpr %3d spill %3d scratch %4d sgpr %3d occ %d' % (
            name, v.get('VGPRs', -1), v.get('AGPRs', -1), v.get('VGPRs Spill', -1), v.get('ScratchSize [bytes/lane]', -1),
            v.get('SGPRs', -1), v.get('Occupancy [waves/SIMD]', -1)))
