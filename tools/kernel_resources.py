#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch report of the HIP library (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py            # compiles mcmc_spec_amd/csrc/msx.hip to /tmp and prints a table
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    for tool in ('c++filt', '/opt/rocm/llvm/bin/llvm-cxxfilt'):
        try:
            out = subprocess.run([tool], input='\n'.join(names), capture_output=True, text=True, check=True).stdout
            return out.strip().split('\n')
        except (OSError, subprocess.CalledProcessError):
            continue
    return names


def main():
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-c', '-mllvm',
           '-amdgpu-kernarg-preload-count=8', '-Rpass-analysis=kernel-resource-usage', '-o', '/tmp/msx_res.o',
           os.path.join(ROOT, 'mcmc_spec_amd', 'csrc', 'msx.hip')] + sys.argv[1:]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = {'name': m.group(1)}
            rows.append(cur)
        for k in ('VGPRs', 'AGPRs', 'SGPRs', 'ScratchSize', 'Occupancy', 'LDS Size'):
            m = re.search(r'remark: .*\b' + k + r'[^:]*: (\d+)', line)
            if m and cur is not None and k not in cur:
                cur[k] = m.group(1)
    names = demangle([r['name'] for r in rows])
    print('{:90s} {:>5s} {:>5s} {:>7s} {:>4s} {:>7s}'.format('kernel', 'VGPR', 'SGPR', 'scratch', 'occ', 'LDS'))
    for r, n in zip(rows, names):
        n = re.sub(r'\(anonymous namespace\)::', '', n)
        n = re.sub(r'\(.*', '', n).replace('void ', '')
        print('{:90s} {:>5s} {:>5s} {:>7s} {:>4s} {:>7s}'.format(n[:90], r.get('VGPRs', '?'), r.get('SGPRs', '?'),
                                                               r.get('ScratchSize', '?'), r.get('Occupancy', '?'),
                                                               r.get('LDS Size', '?')))


if __name__ == '__main__':
    main()
