#!/usr/bin/env python3
"""Config 3 (NADP on the inverted-pendulum model, train_scripts/train_script4mujoco.py:296-411) as a pure ORACLE run: the
reference's optimizer order (optimizer.py:330-362) driven entirely by oracle/mpg_oracle.py - float64 networks and gradients
(nadp.py:87-241 restated), the oracle's analytic cart-pole as the real environment, the oracle's Adam / Polyak - with the
hyper-parameters of tests/test_config34_gpu.py::test_config3_end_to_end_worker_ring_nadp_adam (64 pendulums, batch 512,
replay_starts 3000, sample every 10th iteration, 100-step deterministic evaluation episodes of 16 agents).

Question: the HIP run of this pair reaches returns of -18 .. -24 in 2000 iterations and then DRIFTS in longer
runs (the value mean grows positive although every reward is <= 0).  Is that the algorithm (NADP's bootstrapped target is
unclipped, nadp.py:87-126) or a bug of the device path?  This tool answers it on the CPU: if the float64 oracle shows the same
curve, it is the algorithm.  Test infrastructure (imports oracle/); writes one JSON line per checkpoint.

    python3 tools/config3_oracle_run.py [iterations=6000] [every=250] [seed=0] > profiles/r04_config3_oracle_curve.jsonl"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mpg_oracle as O          # noqa: E402
from tests.golden_inputs import mlp_weights_flat   # noqa: E402


def evaluate(cfg, nets, n_agent, steps, rng):
    env = O.InvertedPendulumContiOracle(n_agent)
    obs = env.reset(rng=rng)
    ret = np.zeros(n_agent)
    th2 = np.zeros(n_agent)
    for _ in range(steps):
        with torch.no_grad():
            a = nets.compute_action(O.process_obses(cfg, torch.as_tensor(obs).to(nets.dtype))).numpy()
        obs, rew, _, _ = env.step(a)
        ret += rew
        th2 += obs[:, 1] ** 2
    return float(ret.mean()), float(np.sqrt(th2 / steps).mean())


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
    every = int(sys.argv[2]) if len(sys.argv) > 2 else 250
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    torch.set_num_threads(4)
    rng = np.random.Generator(np.random.PCG64(seed))
    cfg = O.Cfg(env='InvertedPendulumConti-v0', select=[25], delay_update=1)
    names = ['Q1', 'policy']
    w = {'Q1': mlp_weights_flat(rng, 5, 1), 'policy': mlp_weights_flat(rng, 4, 2)}
    tgt = {k: v.copy() for k, v in w.items()}
    opt = {k: O.AdamState(v.size) for k, v in w.items()}
    num_agent, batch, B, starts = 64, 512, 512, 3000
    env = O.InvertedPendulumContiOracle(num_agent)
    obs = env.reset(rng=rng)
    cap = 500000
    ring = dict(obs=np.zeros((cap, 4), np.float32), act=np.zeros((cap, 1), np.float32))
    size = nxt = 0

    def nets_now(dtype=torch.float64):
        return O.Nets(cfg, w, flat_targets=tgt, dtype=dtype)

    def sample(nets):
        nonlocal obs, size, nxt
        for _ in range(batch // num_agent):
            with torch.no_grad():
                a = nets.compute_action(O.process_obses(cfg, torch.as_tensor(obs).to(nets.dtype))).numpy()
            obs2, rew, done, _ = env.step(a)                       # explore_sigma None (train_script4mujoco.py)
            for i in range(num_agent):
                ring['obs'][nxt], ring['act'][nxt] = obs[i], a[i]
                nxt = (nxt + 1) % cap
                size = min(size + 1, cap)
            # DummyVecEnv semantics: agents that are done are re-drawn, the others continue
            fresh = rng.uniform(-0.01, 0.01, (num_agent, 4))
            env.state = np.where(done[:, None], fresh, env.state)
            obs = env.state.copy()

    nets = nets_now()
    while size < starts:
        sample(nets)
    t0 = time.time()
    for it in range(iters + 1):
        if it % every == 0:
            nets = nets_now()
            ret, th = evaluate(cfg, nets, 16, 100, np.random.Generator(np.random.PCG64(1000 + it)))
            rec = dict(iteration=it, episode_return=round(ret, 3), theta_rms=round(th, 5), wall_s=round(time.time() - t0, 1))
            if it > 0:
                rec.update(value_mean=float(st['value_mean']), q_loss=float(st['q_loss']), policy_loss=float(st['policy_loss']),
                           target_mean=float(np.mean(st['targets'])), target_max=float(np.max(st['targets'])),
                           q_gradient_norm=float(st['q_gradient_norm']), policy_gradient_norm=float(st['policy_gradient_norm']))
            print(json.dumps(rec), flush=True)
        if it == iters:
            break
        nets = nets_now()
        if it % 10 == 0:
            sample(nets)
        idx = rng.integers(0, size, B)
        bt = [ring['obs'][idx], ring['act'][idx]]
        eps_q = rng.standard_normal((cfg.n, B)).astype(np.float32)
        eps_pi = rng.standard_normal((cfg.n, B)).astype(np.float32)
        grads, st = O.nadp_compute_gradient(cfg, nets, bt, eps_q, eps_pi)
        g = {'Q1': np.concatenate([x.ravel() for x in grads[:6]]).astype(np.float32),
             'policy': np.concatenate([x.ravel() for x in grads[6:]]).astype(np.float32)}
        O.apply_gradients(cfg, w, tgt, opt, g, it, names)


if __name__ == '__main__':
    main()
