#!/usr/bin/env python3
"""The device-side twin of tools/config3_oracle_run.py: config 3 (OffPolicyWorker on the analytic cart-pole -> replay ring ->
NADPLearner on the pendulum model -> clip / Adam / Polyak) on the HIP path with the hyper-parameters of
tests/test_config34_gpu.py::test_config3_end_to_end_worker_ring_nadp_adam, one JSON line per checkpoint with the same keys.
    python3 tools/config3_device_run.py [iterations=8000] [every=250] [seed=0] > gpurun_out/config3_device_curve.jsonl"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch   # noqa: E402
from mpg_amd.buffer import ReplayBuffer   # noqa: E402
from mpg_amd.config import default_args   # noqa: E402
from mpg_amd.evaluator import Evaluator   # noqa: E402
from mpg_amd.learners import NADPLearner   # noqa: E402
from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer   # noqa: E402
from mpg_amd.policy import PolicyWithQs   # noqa: E402
from mpg_amd.worker import OffPolicyWorker   # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 250
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
args = default_args('NADP', num_agent=64, batch_size=512, replay_batch_size=512, replay_starts=3000, num_eval_agent=16, fixed_steps=100,
                    seed=seed, init_seed=seed)
worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
learner = NADPLearner(PolicyWithQs, args)
rb = ReplayBuffer(args, 0)
opt = SingleProcessOffPolicyOptimizer(worker, learner, rb, None, args, sampling_interval=10)
ev = Evaluator(PolicyWithQs, args.env_id, args)
ev.share_policy(worker.policy_with_value)
t0 = time.time()
for it in range(0, iters + 1, every):
    m = ev.run_evaluation(it)
    rec = dict(iteration=it, episode_return=round(m['episode_return'], 3), theta_rms=round(m['theta_mse'], 5), wall_s=round(time.time() - t0, 1))
    if it > 0:
        st = learner.get_stats()
        tg = learner.batch_data['batch_targets']
        rec.update({k: float(st[k]) for k in ('value_mean', 'q_loss', 'policy_loss', 'q_gradient_norm', 'policy_gradient_norm')})
        rec.update(target_mean=float(tg.mean().item()), target_max=float(tg.max().item()))
    print(json.dumps(rec), flush=True)
    if it < iters:
        for _ in range(every):
            opt.step()
