#!/usr/bin/env python3
"""Lists kernels whose descriptor asks for the dispatch or queue pointer.  Either means the code reads the AQL packet /
queue descriptor, which live in HOST memory: a scalar load there costs microseconds (round 2: a run-time index into a
4-word array made the compiler move it to LDS and fetch the workgroup size from the packet; 7-30 us on the lanes that draw
the minibatch).  Exit status 1 if any kernel does.   python3 tools/dispatch_ptr_check.py [file.hip ...]"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from mpg_amd import build as B   # noqa: E402
sys.path.insert(0, HERE)
from dpp_hazard_check import asm_of   # noqa: E402


def offenders(files=None):
    files = files or sorted(f for f in os.listdir(B.CSRC) if f.endswith('.hip'))
    bad = []
    for f in files:
        asm = asm_of(f)          # (cached: tests/test_abi.py runs this check and the DPP hazard check on the same dumps)
        for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', asm, re.S):
            what = [k for k in ('dispatch_ptr', 'queue_ptr') if re.search(r'user_sgpr_%s 1' % k, m.group(2))]
            if what:
                name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
                bad.append((f, re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0], what))
    return bad


if __name__ == '__main__':
    bad = offenders(sys.argv[1:])
    for f, n, what in bad:
        print('%s: %s reads %s' % (f, n, ' and '.join(what)))
    print('%d kernel(s) read host-resident packets' % len(bad))
    sys.exit(1 if bad else 0)
