# ISA reading aid: python tools/diag/isa_serial_vmem.py <file.s> [min run]
# Finds SERIALISED global loads: a few global_load* followed by s_waitcnt vmcnt(0), repeated back to back with no matrix instruction or
# barrier between - every wait is a full memory round trip (0.2 - 2 us) that the next load could have been in flight under.
import re, sys
s = open(sys.argv[1]).read()
minrun = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for k in re.split(r'\n(?=_Z[\w]+:\s)', s):
    name = k.split(':')[0]
    if not name.startswith('_Z'): continue
    lines = [l.split(';')[0].strip() for l in k.split('\n')]
    lines = [l for l in lines if l and not l.startswith('.')]
    toks = []
    for i, t in enumerate(lines):
        op = t.split()[0]
        if op.startswith(('global_load', 'scratch_load', 'buffer_load')): toks.append(('R', i))
        elif op == 's_waitcnt' and 'vmcnt(0)' in t: toks.append(('W', i))
        elif op.startswith(('v_mfma', 's_barrier')): toks.append(('X', i))
    string = ''.join(t for t, _ in toks)
    hits = [(m.start(), m.group(0)) for m in re.finditer(r'(?:R{1,6}W){%d,}' % minrun, string)]
    if hits:
        print(name[:130])
        for pos, g in hits:
            i0 = toks[pos][1]
            print('   %d serialised memory round trips from instruction %d (%s): %s' % (g.count('W'), i0, g, lines[i0][:70]))
