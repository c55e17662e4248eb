#!/usr/bin/env python3
"""Kernel durations and the idle gaps between consecutive dispatches from a rocprofv3 --kernel-trace CSV:
    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline
    python3 tools/gaps.py /tmp/tr/*/*kernel_trace.csv
Prints, for the steady-state part of the trace, each kernel's average duration and the average gap in front of it."""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
rows = rows[len(rows) // 3:]                       # steady state
short = lambda n: re.sub(r'\(.*', '', re.sub(r'(void |mlp::|rollout::|\(anonymous namespace\)::)', '', n))[:44]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
for (s0, e0, _), (s1, e1, n1) in zip(rows, rows[1:]):
    k = short(n1)
    dur[k] += e1 - s1
    gap[k] += max(0, s1 - e0)
    cnt[k] += 1
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
print('%-46s %8s %10s %10s' % ('kernel', 'launches', 'avg us', 'gap before us'))
for k in sorted(cnt, key=lambda k: -dur[k]):
    print('%-46s %8d %10.2f %10.2f' % (k, cnt[k], dur[k] / cnt[k] / 1e3, gap[k] / cnt[k] / 1e3))
print('busy %.1f %% of the traced span; total gap %.1f us per 1000 us' % (100.0 * busy / span, 1000.0 * (span - busy) / span))
