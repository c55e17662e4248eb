#!/usr/bin/env python3
"""Trains MPG-v2 with the HIP path at the reference's default learning rates and exports the ONLINE weights for the
trained-weights parity fixture (tests/golden/make_golden.py --only trained_c2):
    python3 tools/train_export.py [iterations=20000]  ->  gpurun_out/trained_weights.npz
Bench configuration (4096 agents, replay batch 4096, sampling every iteration).  Also prints how far the trained networks sit
from the split-fp16 engine's envelope (include/mpg_hip.h): max |parameter| against 1023.5, max first-layer activation
against 4094."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpg_amd.buffer import ReplayBuffer                      # noqa: E402
from mpg_amd.config import default_args                      # noqa: E402
from mpg_amd.evaluator import Evaluator                      # noqa: E402
from mpg_amd.learners import MPGLearner                      # noqa: E402
from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer   # noqa: E402
from mpg_amd.policy import PolicyWithQs                      # noqa: E402
from mpg_amd.worker import OffPolicyWorker                   # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
B = 4096
args = default_args('MPG-v2', num_agent=B, batch_size=B, replay_batch_size=B, replay_starts=4 * B, num_eval_agent=256)
worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
opt = SingleProcessOffPolicyOptimizer(worker, MPGLearner(PolicyWithQs, args), ReplayBuffer(args, 0), None, args, sampling_interval=1)
ev = Evaluator(PolicyWithQs, args.env_id, args)
ev.share_policy(worker.policy_with_value)
pw = worker.policy_with_value
print(json.dumps(dict(iteration=0, episode_return=round(ev.run_evaluation(0)['episode_return'], 2))))
for _ in range(iters):
    opt.step()
print(json.dumps(dict(iteration=iters, episode_return=round(ev.run_evaluation(iters)['episode_return'], 2))))
pw.check_status()                              # raises if the engine left its envelope anywhere along the run
assert int(pw.nonfinite.sum().item()) == 0
out = {}
for n in pw.names:
    out['w_' + n] = pw.net(n).cpu().numpy()
    w = pw._as_list(pw.net(n), n)
    print(n, 'max|W1| %.3f max|W2| %.3f max|W3| %.3f' % tuple(float(w[i].abs().max()) for i in (0, 2, 4)))
# first-layer activations on a reset-law batch
obs = worker.env.reset()
x = obs * torch.tensor([pw.cfg.obs_scale[i] for i in range(6)], device=obs.device)
wp = pw._as_list(pw.net('policy'), 'policy')
h1 = torch.nn.functional.elu(x @ wp[0] + wp[1])
print('policy: max first-layer activation %.2f (envelope 4094)' % float(h1.max()))
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, 'gpurun_out', 'trained_weights.npz'), iterations=iters, **out)
print('wrote gpurun_out/trained_weights.npz')
