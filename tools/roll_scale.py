#!/usr/bin/env python3
"""Per-step and fixed cost of the two rollout sweeps: times the bench workload's gradient step with horizons n = 5, 15, 25
(kernel timer slots 0 / 1) and fits  t(n) = fixed + (n + 1) * per_step.   python3 tools/roll_scale.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpg_amd import ops                                      # noqa: E402
from mpg_amd.buffer import ReplayBuffer                      # noqa: E402
from mpg_amd.config import default_args                      # noqa: E402
from mpg_amd.learners import MPGLearner                      # noqa: E402
from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer, quiesce_gc   # noqa: E402
from mpg_amd.policy import PolicyWithQs                      # noqa: E402
from mpg_amd.worker import OffPolicyWorker                   # noqa: E402

B = int(os.environ.get("MPG_B", "4096"))
res = {}
for n in (5, 15, 25):
    args = default_args('MPG-v2', num_agent=B, batch_size=B, replay_batch_size=B, replay_starts=4 * B, max_buffer_size=500000,
                        num_rollout_list_for_policy_update=[0, n])
    worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, worker_id=0)
    opt = SingleProcessOffPolicyOptimizer(worker, MPGLearner(PolicyWithQs, args), ReplayBuffer(args, 0), None, args, sampling_interval=1)
    prof = ops.Profiler(max_samples=64)
    opt.set_profiler(prof)
    quiesce_gc()
    for _ in range(150):
        opt.step()
    torch.cuda.synchronize()
    prof.start(8)
    for _ in range(320):
        opt.step()
    torch.cuda.synchronize()
    res[n] = (prof.read(0)[0] * 1e3, prof.read(1)[0] * 1e3)
    prof.stop()
    opt.set_profiler(None)
for k, name in ((0, 'fwd'), (1, 'bwd')):
    ns = np.array(sorted(res))
    t = np.array([res[n][k] for n in ns])
    per, fixed = np.polyfit(ns + 1, t, 1)
    print('%s: %s us -> per step %.3f us, fixed %.2f us' % (name, ' '.join('n=%d %.1f' % (n, x) for n, x in zip(ns, t)), per, fixed))
