#!/usr/bin/env python3
"""Device assembly of one translation unit as the shipped flags compile it (no GPU needed):
    python3 tools/asm.py rollout_fwd.hip [extra flags ...]  ->  scratch_asm/<file>.s   (scratch_asm/ is git-ignored)"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from mpg_amd import build as B   # noqa: E402

f = sys.argv[1]
out = os.path.join(HERE, '..', 'scratch_asm', f + '.s')
os.makedirs(os.path.dirname(out), exist_ok=True)
flags = B.COMMON + B.EXTRA.get(f, []) + sys.argv[2:] + ['-x', 'hip', '--offload-device-only', '-S']
subprocess.check_call([B.hipcc()] + flags + [os.path.join(B.CSRC, f), '-o', out])
print(os.path.abspath(out))
