# ISA reading aid: python tools/diag/isa_serial_lds.py <file.s> [min run]
# Finds SERIALISED LDS reads: one or two ds_read* followed by s_waitcnt lgkmcnt(0), repeated back to back - each wait is a full LDS
# round trip (~64 - 130 cycles) on the wave that executes it.  The usual cause is a destination register range that overlaps the
# previous read's (a partly dead b128): write-after-write, so the compiler must drain before the next read (EXPERIMENTS.md section 5.10).
import re, sys
s = open(sys.argv[1]).read()
minrun = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for k in re.split(r'\n(?=_Z[\w]+:\s)', s):
    name = k.split(':')[0]
    if not name.startswith('_Z'): continue
    lines = [l.split(';')[0].strip() for l in k.split('\n')]
    lines = [l for l in lines if l and not l.startswith('.')]
    toks = []
    for i, t in enumerate(lines):
        op = t.split()[0]
        if op.startswith('ds_read'): toks.append(('R', i))
        elif op == 's_waitcnt' and 'lgkmcnt(0)' in t: toks.append(('W', i))
        elif op.startswith(('v_mfma', 's_barrier', 's_cbranch', 's_branch', 'ds_write')): toks.append(('X', i))
    string = ''.join(t for t, _ in toks)
    hits = [(m.start(), m.group(0)) for m in re.finditer(r'(?:R{1,2}W){%d,}' % minrun, string)]
    if hits:
        print(name[:130])
        for pos, g in hits:
            i0 = toks[pos][1]
            print('   %d serialised round trips from instruction %d: %s' % (g.count('W'), i0, lines[i0][:80]))
