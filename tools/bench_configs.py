"""Side measurements for DESIGN.md (NOT the bench line - bench.py measures BASELINE.json's metric on config C2):
  * env kernel: step-only and step+reset rates (SURVEY.md §8d asks for both), at 4096 agents (latency-bound) and at a
    size that fills the chip;
  * gradient-step time of the other BASELINE configs: C1 (MPG-v2, B=256), C3 (NADP on the pendulum model, B=8192),
    C4 (TD3 + prioritized replay, B=65 536, tree capacity 2^19), each = replay/compute_gradient/apply_gradients.
Prints one JSON object per line.  Run on the GPU box:  python tools/bench_configs.py
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

DEV = 'cuda'


def timed(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def env_rates():
    from mpg_amd.envs import PathTrackingEnv
    for n in (4096, 65536, 1 << 20):
        env = PathTrackingEnv(num_agent=n, device=DEV, seed=1)
        env.reset()
        act = torch.rand(n, 2, device=DEV) * 2 - 1
        t_step = timed(lambda: env.step(act), 50)

        def step_reset():
            env.step(act)
            env.reset()
        t_sr = timed(step_reset, 50)
        print(json.dumps(dict(what='env', num_agent=n, step_only_us=t_step * 1e6, step_only_env_steps_per_s=n / t_step,
                              step_reset_us=t_sr * 1e6, step_reset_env_steps_per_s=n / t_sr,
                              hbm_bytes_per_env_step=85, hbm_frac_step_only=85 * n / t_step / 8e12)))


def c1():
    from mpg_amd.buffer import ReplayBuffer
    from mpg_amd.config import default_args
    from mpg_amd.learners import MPGLearner
    from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer
    from mpg_amd.policy import PolicyWithQs
    from mpg_amd.worker import OffPolicyWorker
    args = default_args('MPG-v2')                       # reference defaults: 8 agents, 512 per sample, B = 256
    worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
    opt = SingleProcessOffPolicyOptimizer(worker, MPGLearner(PolicyWithQs, args), ReplayBuffer(args, 0), None, args)
    t = timed(opt.step, 500, 50)
    print(json.dumps(dict(what='C1 MPG-v2 B=256, 8 agents x 64 env steps every 10th iteration (reference defaults)',
                          ms_per_iteration=t * 1e3, grad_steps_per_s=1 / t, env_steps_per_s=51.2 / t)))


def c3():
    from mpg_amd.config import default_args
    from mpg_amd.learners import NADPLearner
    from mpg_amd.policy import PolicyWithQs
    B = 8192
    args = default_args('NADP', replay_batch_size=B)
    learner = NADPLearner(PolicyWithQs, args)
    pw = learner.policy_with_value
    g = torch.Generator(device='cpu').manual_seed(0)
    obs = (torch.randn(B, 4, generator=g) * torch.tensor([0.5, 0.1, 0.5, 0.5])).to(DEV)
    act = ((torch.rand(B, 1, generator=g) * 6) - 3).to(DEV)
    batch = [obs, act, torch.zeros(B, device=DEV), obs, torch.zeros(B, device=DEV)]
    it = [0]

    def step():
        learner.compute_gradient(batch, None, None, it[0])
        pw.apply_gradients(it[0], learner.flat_grad)
        it[0] += 1
    t = timed(step, 100, 10)
    flop = 14.8e6 * B
    print(json.dumps(dict(what='C3 NADP pendulum model B=8192 (compute_gradient + apply_gradients)', ms_per_grad_step=t * 1e3,
                          grad_steps_per_s=1 / t, algorithmic_gflop=flop / 1e9, algorithmic_tflops=flop / t / 1e12, executed_f16_tflops=3 * flop / t / 1e12,
                          frac_f16_mfma_peak=3 * flop / t / 2500e12)))


def c4():
    from mpg_amd.buffer import PrioritizedReplayBuffer
    from mpg_amd.config import default_args
    from mpg_amd.learners import TD3Learner
    from mpg_amd.policy import PolicyWithQs
    from tests.golden_inputs import reset_law_obs
    B, N = 65536, 500000
    args = default_args('TD3', replay_batch_size=B, buffer_type='priority', replay_starts=N)
    learner = TD3Learner(PolicyWithQs, args)
    pw = learner.policy_with_value
    rb = PrioritizedReplayBuffer(args, 0)
    rng = np.random.Generator(np.random.PCG64(0))
    chunk = 50000
    for _ in range(N // chunk):
        rb.add_batch((torch.as_tensor(reset_law_obs(rng, chunk)).to(DEV), torch.as_tensor(rng.uniform(-1, 1, (chunk, 2)), dtype=torch.float32).to(DEV),
                      torch.as_tensor(rng.standard_normal(chunk), dtype=torch.float32).to(DEV),
                      torch.as_tensor(reset_law_obs(rng, chunk)).to(DEV), torch.ones(chunk, dtype=torch.uint8, device=DEV)))
    it = [0]

    def step():
        s = rb.replay()
        learner.compute_gradient(s[:5], rb, s[-1], it[0])
        info = learner.get_info_for_buffer()
        info['rb'].update_priorities(info['indexes'], info['td_error'])
        pw.apply_gradients(it[0], learner.flat_grad)
        it[0] += 1
    t = timed(step, 50, 5)
    t_per = timed(lambda: rb.replay(), 50, 5)
    flop = 2.17e6 * B
    print(json.dumps(dict(what='C4 TD3 + prioritized replay B=65536, 500k transitions (replay + compute_gradient + update_priorities + apply_gradients)',
                          ms_per_grad_step=t * 1e3, grad_steps_per_s=1 / t, replay_rows_per_s=B / t, per_sample_gather_ms=t_per * 1e3,
                          algorithmic_gflop=flop / 1e9, algorithmic_tflops=flop / t / 1e12, executed_f16_tflops=3 * flop / t / 1e12,
                          frac_f16_mfma_peak=3 * flop / t / 2500e12)))


if __name__ == '__main__':
    torch.zeros(1, device=DEV)
    which = sys.argv[1:] or ['env_rates', 'c1', 'c3', 'c4']
    for name in which:
        globals()[name]()
