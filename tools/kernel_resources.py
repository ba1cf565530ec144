#!/usr/bin/env python3
"""VGPRs / spills / LDS / occupancy of every kernel of one csrc file, as the compiler reports them
(hipcc -Rpass-analysis=kernel-resource-usage).  Usage: python tools/kernel_resources.py kernels_walk.hip [filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = os.path.join(ROOT, 'n2nmn_amd', 'csrc', sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-x', 'hip', '-c', src,
           '-o', '/tmp/_kr.o', '-Rpass-analysis=kernel-resource-usage']
    out = subprocess.run(cmd, capture_output=True, text=True, cwd='/tmp').stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = subprocess.run(['c++filt', m.group(1)], capture_output=True,
                                 text=True).stdout.strip()
            cur = re.sub(r'\(n2nmn::ModuleW.*|\(n2nmn::WalkA.*', '', cur).replace('n2nmn::(anonymous namespace)::', '').replace('void ', '')
            rows[cur] = {}
            continue
        m = re.search(r'remark: +([A-Za-z ]+\w)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\d+)', line)
        if m and cur:
            rows[cur][m.group(1).strip()] = int(m.group(2))
    print('%-44s %5s %5s %6s %7s %5s %7s' % ('kernel', 'VGPR', 'AGPR', 'spill', 'scratch', 'occ', 'LDS'))
    for k, r in rows.items():
        if flt in k:
            print('%-44s %5d %5d %6d %7d %5d %7d' % (k[:44], r.get('VGPRs', -1), r.get('AGPRs', -1),
                                                   r.get('VGPRs Spill', -1), r.get('ScratchSize', -1),
                                                   r.get('Occupancy', -1), r.get('LDS Size', -1)))


if __name__ == '__main__':
    main()
