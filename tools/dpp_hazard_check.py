#!/usr/bin/env python3
"""DPP read-after-VALU-write hazard check on the shipped device code.

gfx9 / CDNA: a DPP instruction that reads, through its DPP operand (src0), a VGPR written by a VALU instruction needs TWO wait
states in between (any two instructions, or `s_nop 1`).  The compiler inserts them for its own instructions but does not look
inside inline asm: the hand-written `v_add_f32_dpp` reduction stages of mlp_core.h rely on their stage-major order (and one
`s_nop 1` ahead of the first stage) to keep every DPP read behind the write of its operand.  This script checks that property
on the ISA the shipped flags produce, for every kernel of every translation unit (no GPU needed): exit status 1 and one line
per violation.    python3 tools/dpp_hazard_check.py [file.hip ...]"""
import functools
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from mpg_amd import build as B   # noqa: E402

NEED = 2
REG = re.compile(r'^v(\d+)$|^v\[(\d+):(\d+)\]$')


def regs(op):
    m = REG.match(op.strip())
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


@functools.lru_cache(maxsize=None)
def asm_of(f):
    flags = B.COMMON + B.EXTRA.get(f, []) + ['-x', 'hip', '--offload-device-only', '-S']
    return subprocess.run([B.hipcc()] + flags + [os.path.join(B.CSRC, f), '-o', '-'], capture_output=True, text=True).stdout


def violations(asm, fname=''):
    bad, n_dpp = [], 0
    func = None
    hist = []          # most recent first: (written VGPRs, wait states this instruction provides, text)
    for line in asm.splitlines():
        t = line.split(';')[0].strip()
        if not t:
            continue
        if t.endswith(':'):
            if not t.startswith('.'):
                func = t[:-1]
            hist = []                                   # a label: nothing is known about what ran before
            continue
        if t.startswith('.'):
            continue
        parts = t.split(None, 1)
        op = parts[0]
        args = [a.strip() for a in parts[1].split(',')] if len(parts) > 1 else []
        is_dpp = '_dpp' in op or ' quad_perm:' in t or ' row_' in t or ' wave_' in t
        if is_dpp and len(args) >= 2:
            n_dpp += 1
            src0 = regs(args[1].split()[0])
            ws = 0
            for wr, w, txt in hist:
                if ws >= NEED:
                    break
                if wr & src0:
                    bad.append((fname, func, t, txt, ws))
                    break
                ws += w
        written = set()
        if op.startswith('v_') and args and not op.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')):
            written = regs(args[0].split()[0])
        w = 1
        if op == 's_nop':
            w = int(args[0], 0) + 1
        hist.insert(0, (written, w, t))
        del hist[8:]
    return bad, n_dpp


def check(files=None):
    files = files or sorted(f for f in os.listdir(B.CSRC) if f.endswith('.hip'))
    bad, n = [], 0
    for f in files:
        b, k = violations(asm_of(f), f)
        bad += b
        n += k
    return bad, n


if __name__ == '__main__':
    bad, n = check(sys.argv[1:])
    for f, func, t, prod, ws in bad:
        name = subprocess.run(['c++filt', func or ''], capture_output=True, text=True).stdout.strip().split('(')[0]
        print('%s: %s: "%s" reads a register written %d wait state(s) earlier by "%s"' % (f, name, t, ws, prod))
    print('%d DPP instruction(s) checked, %d hazard(s)' % (n, len(bad)))
    sys.exit(1 if bad else 0)
