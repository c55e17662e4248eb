#!/usr/bin/env python3
"""Measured errors of the MPG-v1 path (25 real-env steps + n-step target + gradients) against the reference fixture:
what the tolerances in tests/test_learner_gpu.py / test_networks_gpu.py are derived from.  GPU box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mpg_oracle as O   # noqa: E402  (tool, not product)


def rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    from mpg_amd import ops
    from mpg_amd.config import default_args
    from mpg_amd.envs import PathTrackingEnv
    from mpg_amd.learners import MPGLearner
    from mpg_amd.policy import PolicyWithQs
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'mpg_v1_H256_B64.npz')))
    dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).cuda()
    args = default_args('MPG-v1', replay_batch_size=64, num_batch_reuse=1)
    learner = MPGLearner(PolicyWithQs, args)
    pw = learner.policy_with_value
    flat = np.concatenate([g['w_' + n] for n in pw.names])
    pw.set_flat(flat, (flat * np.float32(g['target_scale'])).astype(np.float32))
    B = 64
    env = PathTrackingEnv(num_agent=B)
    obs = dev(g['batch_obs'])
    env.reset(init_obs=obs)
    rewards, obs_l = [], []
    for t in range(25):
        a = dev(g['batch_actions']) if t == 0 else ops.policy_action(pw.cfg, pw.net('policy'), obs)
        obs, r, _, _ = env.step(a)
        rewards.append(r)
        obs_l.append(obs.clone())
    rw = torch.stack(rewards).cpu().numpy()
    ref = g['nstep_all_rewards']
    for t in (0, 1, 5, 12, 24):
        print('step %2d reward: max abs err %.3e  max rel %.3e   last-obs max abs err %s' % (
            t, np.abs(rw[t] - ref[t]).max(), (np.abs(rw[t] - ref[t]) / np.maximum(np.abs(ref[t]), 1e-3)).max(),
            '%.3e' % np.abs(obs_l[t].cpu().numpy() - g['nstep_last_obs']).max() if t == 24 else '-'))
    worst = np.unravel_index(np.argmax(np.abs(rw - ref)), rw.shape)
    print('worst reward element', worst, rw[worst], ref[worst])
    batch = [dev(g[k]) for k in ('batch_obs', 'batch_actions', 'batch_rewards', 'batch_obs_tp1', 'batch_dones')]
    learner.counter = 0
    grads = learner.compute_gradient(batch, None, None, 100, eps=dev(g['eps']))
    got = torch.cat([x.reshape(-1) for x in grads]).cpu().numpy()
    y = learner.batch_data['batch_targets'].cpu().numpy()
    print('targets: rel-l2 %.3e  max abs %.3e (max |y| %.3f);  reference f32-vs-f64 rel-l2 %.3e' % (
        rel(y, g['it100_targets']), np.abs(y - g['it100_targets']).max(), np.abs(g['it100_targets']).max(),
        rel(g['it100_targets'], g['it100_targets_f64'])))
    o = 0
    for name in pw.names:
        din, dout = pw.dims[name]
        for shp in O.mlp_shapes(din, 256, dout):
            n = int(np.prod(shp))
            if np.linalg.norm(g['it100_grads'][o:o + n]) > 0:
                print('grad %-6s %-10s rel-l2 %.3e' % (name, shp, rel(got[o:o + n], g['it100_grads'][o:o + n])))
            o += n


if __name__ == '__main__':
    main()
