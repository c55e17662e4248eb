#!/usr/bin/env python3
"""Step time of the bench workload with look-ahead observations (num_future_data = K): the paths the bench itself never times - 16-/24-wide
network kernels, `WIDE` sweeps (which carry spilled registers), launch-per-stage gradients.  python3 tools/lookahead_step_time.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpg_amd.buffer import ReplayBuffer
from mpg_amd.config import default_args
from mpg_amd.learners import MPGLearner
from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer, quiesce_gc
from mpg_amd.policy import PolicyWithQs
from mpg_amd.worker import OffPolicyWorker

B = 4096
for K in (0, 3, 10):
    args = default_args('MPG-v2', num_agent=B, batch_size=B, replay_batch_size=B, replay_starts=4 * B, max_buffer_size=500000, num_future_data=K)
    worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
    opt = SingleProcessOffPolicyOptimizer(worker, MPGLearner(PolicyWithQs, args), ReplayBuffer(args, 0), None, args, sampling_interval=1)
    quiesce_gc()
    for _ in range(150):
        opt.step()
    regions = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            opt.step()
        torch.cuda.synchronize()
        regions.append((time.perf_counter() - t0) / 100 * 1e3)
    worker.policy_with_value.check_status()
    print(json.dumps({'num_future_data': K, 'obs_dim': 6 + K, 'rows': B, 'native_driver': opt._fused is not None,
                      'ms_per_step_median': sorted(regions)[2], 'regions': [round(r, 4) for r in regions]}), flush=True)
    del opt, worker
