#!/usr/bin/env python3
"""Register / spill / LDS usage of every kernel as the shipped flags compile it (cross-compiles without a GPU):
    python3 tools/kernel_resources.py [file.hip ...]        (hipcc -Rpass-analysis=kernel-resource-usage)"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from mpg_amd import build as B   # noqa: E402

files = sys.argv[1:] or ['rollout_fwd.hip', 'rollout_bwd.hip', 'fused_kernels.hip', 'mlp_kernels.hip']
for f in files:
    flags = B.COMMON + B.EXTRA.get(f, []) + ['-x', 'hip', '--offload-device-only', '-Rpass-analysis=kernel-resource-usage']
    out = subprocess.run([B.hipcc()] + flags + ['-c', os.path.join(B.CSRC, f), '-o', '/dev/null'], capture_output=True, text=True).stderr
    cur = None
    for line in out.splitlines():
        m = re.search(r'remark: (?:[^:]+:\d+:\d+: )?\s*(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (.*?)\s*(\[-R.*)?$', line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == 'Function Name':
            if cur:
                print(cur)
            name = subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip()
            name = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0].replace('void ', '')
            cur = '%-46s' % name
        else:
            short = {'VGPRs': 'vgpr', 'AGPRs': 'agpr', 'ScratchSize [bytes/lane]': 'scratch', 'Occupancy [waves/SIMD]': 'occ',
                     'SGPRs Spill': 'sspill', 'VGPRs Spill': 'vspill', 'LDS Size [bytes/block]': 'lds'}[k]
            cur += ' %s=%s' % (short, v)
    if cur:
        print(cur)
