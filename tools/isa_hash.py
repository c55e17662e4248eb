#!/usr/bin/env python3
"""Hash of the DEVICE assembly of every translation unit as the shipped flags compile it, per engine (no GPU needed): the check that a
source clean-up changed no instruction.   python3 tools/isa_hash.py > before.txt; ...edit...; python3 tools/isa_hash.py | diff before.txt -"""
import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from mpg_amd import build as B   # noqa: E402


def one(job):
    engine, f, side = job
    vflags = B.VARIANTS[engine][2]
    how = ['-x', 'hip', '--offload-device-only'] if side == 'device' else (['-x', 'hip', '--offload-host-only'] if f.endswith('.hip') else [])
    flags = B.COMMON + vflags + B.EXTRA.get(f, []) + how + ['-S', '-o', '-']
    asm = subprocess.run([B.hipcc()] + flags + [os.path.join(B.CSRC, f)], capture_output=True, text=True, check=True).stdout
    keep = []
    cm = ';' if side == 'device' else '#'
    for ln in asm.split('\n'):
        ln = ln.split(cm)[0].rstrip()                   # comments (source line echoes, resource summaries) are not code
        if not ln or re.match(r'^\s*\.(file|loc|ident|section\s+\.debug|cfi_)', ln):
            continue
        keep.append(ln)
    n = sum(1 for k in keep if re.match(r'^\s+[sv]_', k)) if side == 'device' else sum(1 for k in keep if re.match(r'^\s+[a-z]', k))
    return engine, f, side, hashlib.sha256('\n'.join(keep).encode()).hexdigest()[:16], n


if __name__ == '__main__':
    jobs = [(e, f, 'device') for e in B.VARIANTS for f in B.sources() if f.endswith('.hip')]
    jobs += [(e, f, 'host') for e in B.VARIANTS for f in B.sources()]        # (embedded device images are left out by --offload-host-only)
    with ThreadPoolExecutor(8) as ex:
        for engine, f, side, h, n in ex.map(one, jobs):
            print('%-6s %-6s %-28s %s  %6d instructions' % (engine, side, f, h, n))
