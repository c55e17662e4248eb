#!/usr/bin/env python3
"""Partial preprocessor (round 6): resolves every conditional on the EXPERIMENT macros of mpg_amd/csrc - all
MPG_AB_* (undefined), MPG_TR_IMAGE, MPG_IMG_INTERLEAVE (undefined), MPG_IMG_LAYOUT (0) - and leaves every other conditional
(MPG_F32_MFMA / MPG_SPLIT, MPG_STAMP, MPG_TIMELINE, include guards ...) exactly as it is, so that the shipped sources hold the shipped
expansion only.  The removed branches live on as a patch: `git apply archive/proto/ablation_macros.patch` puts the scaffolding back
(the A/B scripts under tools/ do that first).

    python3 tools/strip_ablation.py            # rewrites mpg_amd/csrc/* in place
    python3 tools/strip_ablation.py --check    # exits 1 if any experiment macro is left

Conditions are handled in the forms the sources use: #ifdef / #ifndef X;  #if / #elif of `defined(X)` / `!defined(X)` terms joined by
&& or by || (never mixed);  #if / #elif MPG_IMG_LAYOUT == n."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'mpg_amd', 'csrc')
UNDEF = re.compile(r'^(MPG_AB_[A-Z0-9_]+|MPG_TR_IMAGE|MPG_IMG_INTERLEAVE)$')
VALUE = {'MPG_IMG_LAYOUT': 0}
EXPERIMENT = re.compile(r'MPG_AB_[A-Z0-9_]+|MPG_TR_IMAGE|MPG_IMG_INTERLEAVE|MPG_IMG_LAYOUT')


def term(t):
    """-> True / False / the term itself (unknown)"""
    t = t.strip()
    m = re.match(r'^(!?)\s*defined\s*\(\s*(\w+)\s*\)$', t)
    if m:
        neg, name = m.group(1) == '!', m.group(2)
        if UNDEF.match(name):
            return neg
        if name in VALUE:
            return not neg
        return t
    m = re.match(r'^(\w+)\s*==\s*(\d+)$', t)
    if m and m.group(1) in VALUE:
        return VALUE[m.group(1)] == int(m.group(2))
    assert not EXPERIMENT.search(t), 'unhandled condition term: %r' % t
    return t


def evaluate(expr):
    """-> True / False / simplified expression text"""
    expr = expr.split('//')[0].strip()
    if '&&' in expr and '||' in expr:
        assert not EXPERIMENT.search(expr), 'mixed && / || on an experiment macro: %r' % expr
        return expr
    if '||' in expr:
        ts = [term(t) for t in expr.split('||')]
        if any(t is True for t in ts):
            return True
        ts = [t for t in ts if t is not False]
        return ' || '.join(ts) if ts else False
    ts = [term(t) for t in expr.split('&&')]
    if any(t is False for t in ts):
        return False
    ts = [t for t in ts if t is not True]
    return ' && '.join(ts) if ts else True


def strip(text):
    out = []
    # stack entries: dict(kind='keep' | 'resolved', emitting=bool, taken=bool, parent_emit=bool)
    stack = []

    def emitting():
        return all(f['emit'] for f in stack)
    for line in text.split('\n'):
        s = line.strip()
        m = re.match(r'^#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)$', s)
        if not m:
            if emitting():
                out.append(line)
            continue
        d, rest = m.group(1), m.group(2)
        comment = ''
        if d in ('ifdef', 'ifndef', 'if'):
            if d == 'if':
                v = evaluate(rest)
            else:
                name = rest.split('//')[0].strip()
                v = term(('!' if d == 'ifndef' else '') + 'defined(' + name + ')')
                if isinstance(v, str):
                    v = None                       # unknown #ifdef / #ifndef: keep the line as written
            parent = emitting()
            if v is True or v is False:
                stack.append(dict(kind='resolved', emit=v, taken=v))
            else:
                stack.append(dict(kind='keep', emit=True, taken=False))
                if parent:
                    if d == 'if' and isinstance(v, str) and v != rest.split('//')[0].strip():
                        tail = rest.split('//', 1)
                        comment = ('   //' + tail[1]) if len(tail) > 1 else ''
                        out.append(line[:len(line) - len(line.lstrip())] + '#if ' + v + comment)
                    else:
                        out.append(line)
            continue
        f = stack[-1]
        if d == 'elif':
            if f['kind'] == 'keep':
                v = evaluate(rest)
                assert isinstance(v, str) and v == rest.split('//')[0].strip(), 'experiment macro in an #elif of a kept chain: %r' % line
                if all(g['emit'] for g in stack[:-1]):
                    out.append(line)
            else:
                if f['taken']:
                    f['emit'] = False
                else:
                    v = evaluate(rest)
                    assert v is True or v is False, 'partly known #elif chain: %r' % line
                    f['emit'] = f['taken'] = v
        elif d == 'else':
            if f['kind'] == 'keep':
                if all(g['emit'] for g in stack[:-1]):
                    out.append(line)
            else:
                f['emit'] = not f['taken']
                f['taken'] = True
        else:   # endif
            stack.pop()
            if f['kind'] == 'keep' and emitting():
                out.append(line)
    assert not stack
    return '\n'.join(out)


def main():
    check = '--check' in sys.argv
    left = 0
    for path in sorted(glob.glob(os.path.join(CSRC, '*'))):
        if not os.path.isfile(path):
            continue
        text = open(path).read()
        if check:
            # mentions in comments are history, not scaffolding: only preprocessor lines count
            n = sum(1 for ln in text.split('\n') if ln.strip().startswith('#') and EXPERIMENT.search(ln.split('//')[0]))
            if n:
                print('%s: %d conditional(s) on experiment macros' % (os.path.relpath(path, ROOT), n))
            left += n
            continue
        new = strip(text)
        if new != text:
            open(path, 'w').write(new)
            print('stripped', os.path.relpath(path, ROOT))
    if check:
        sys.exit(1 if left else 0)


if __name__ == '__main__':
    main()
