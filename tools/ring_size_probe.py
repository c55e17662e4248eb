#!/usr/bin/env python3
"""Does the minibatch gather (inside k_target_fused) depend on the ring size?  Times the native step for several replay
capacities.  python3 tools/ring_size_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402


def run(cap):
    from mpg_amd import ops
    from mpg_amd.buffer import ReplayBuffer
    from mpg_amd.config import default_args
    from mpg_amd.learners import MPGLearner
    from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer, quiesce_gc
    from mpg_amd.policy import PolicyWithQs
    from mpg_amd.worker import OffPolicyWorker
    B = bench.B_PER_GPU
    args = default_args('MPG-v2', num_agent=B, batch_size=B, replay_batch_size=B, replay_starts=4 * B, max_buffer_size=cap, seed=0)
    worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, worker_id=0)
    learner = MPGLearner(PolicyWithQs, args)
    opt = SingleProcessOffPolicyOptimizer(worker, learner, ReplayBuffer(args, 0), None, args, sampling_interval=1)
    prof = ops.Profiler(max_samples=200)
    opt.set_profiler(prof)
    quiesce_gc()
    for _ in range(300):
        opt.step()
    torch.cuda.synchronize()
    prof.start(4)
    t0 = time.perf_counter()
    for _ in range(400):
        opt.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('capacity %7d: %.4f ms/step   k_target_fused %.4f ms   k_critic_fused %.4f ms' % (cap, 1e3 * dt / 400, prof.read(6)[0], prof.read(7)[0]))


if __name__ == '__main__':
    for cap in (16384, 65536, 262144, 500000, 2000000):
        run(cap)
