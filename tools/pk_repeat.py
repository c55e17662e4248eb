#!/usr/bin/env python3
"""Launches the bench-size MPG-v2 gradient (c2_mpg_v2_B4096) N times on identical inputs and prints how many launches differ
from the first one bit for bit, and where (tools/pk_anomaly.sh).   python3 tools/pk_repeat.py [N=5000]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpg_amd.config import default_args                      # noqa: E402
from mpg_amd.learners import MPGLearner                      # noqa: E402
from mpg_amd.policy import PolicyWithQs                      # noqa: E402
from tests import yardstick as Y                             # noqa: E402
from tests.golden_inputs import bench_case_inputs            # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).cuda()
d = bench_case_inputs('c2_mpg_v2_B4096')
flat = {k: np.concatenate([np.asarray(w).ravel() for w in v]).astype(np.float32) for k, v in d['nets'].items()}
batch = [dev(x) for x in d['batch']]
learner = MPGLearner(PolicyWithQs, default_args('MPG-v2', replay_batch_size=d['B'], num_batch_reuse=1))
pw = learner.policy_with_value
w = np.concatenate([flat[n] for n in pw.names])
pw.set_flat(w, (w * np.float32(0.97)).astype(np.float32))
eps = dev(d['eps'])


def run():
    learner.counter = 0
    return torch.cat([x.reshape(-1) for x in learner.compute_gradient(batch, None, None, 100, eps=eps)]).clone()


# reference result: the SHIPPED build's gradient (every multiply-add one instruction; bit-identical over 20 000 launches),
# saved by the first variant of tools/pk_anomaly.sh - the packed form performs the same fmas on the same operands, so it
# must reproduce it bit for bit
REF = os.path.join(ROOT, 'gpurun_out', 'pk_anomaly', 'ref_shipped.pt')
first = run()
if os.environ.get('PK_SAVE_REF'):
    os.makedirs(os.path.dirname(REF), exist_ok=True)
    torch.save(first.cpu(), REF)
ref = torch.load(REF).cuda() if os.path.exists(REF) else first
lay, _ = Y.layout([(n,) + tuple(pw.dims[n]) for n in pw.names])
bad, where, outcomes, sizes = 0, {}, {}, []
for k in range(N):
    g = run()
    key = hash(g.cpu().numpy().tobytes()) if k < 400 else None         # distinct outcomes among the first 400 launches
    if key is not None:
        outcomes[key] = outcomes.get(key, 0) + 1
    if not torch.equal(g, ref):
        bad += 1
        idx = torch.nonzero(g != ref).flatten().cpu().numpy()
        sizes.append(idx.size)
        for name, shp, o, n in lay:
            c = int(((idx >= o) & (idx < o + n)).sum())
            if c:
                where['%s %s' % (name, shp)] = where.get('%s %s' % (name, shp), 0) + 1
print('%d of %d launches differ from the shipped build\'s result (%.2f %%); %d distinct results among the first %d launches'
      % (bad, N, 100.0 * bad / N, len(outcomes), min(N, 400)))
if sizes:
    print('   elements that differ per bad launch: min %d, median %d, max %d' % (min(sizes), sorted(sizes)[len(sizes) // 2], max(sizes)))
for k, v in sorted(where.items(), key=lambda kv: -kv[1]):
    print('   %-22s differs in %d launches' % (k, v))
