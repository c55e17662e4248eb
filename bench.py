#!/usr/bin/env python3
"""bench.py - headline benchmark of the MPG hot path on MI355X.

Workload (BASELINE.json configs[1]): PathTrackingEnv, MPG (learner_version MPG-v2, the reference script's default,
train_script.py:847), n = 25, M = 1, 4096 vectorised envs and a replay batch of 4096 PER GPU.
One "step" = one pass of the hot path over one batch, in SingleProcessOffPolicyOptimizer.step order
(optimizer.py:330-362): worker.sample (policy + N(0, .1) noise -> env.step (20 sub-steps) -> env.reset, 4096 agents)
-> replay_buffer.add_batch -> replay_buffer.replay (4096 rows) -> learner.compute_gradient (clipped double-Q target,
Q1/Q2 loss+grad, 25-step model rollout + mixed policy gradient) -> all-reduce (N > 1) -> clip_by_global_norm ->
worker.apply_gradients (Adam + Polyak).  So every step is 4096 env-steps and 1 gradient step per GPU.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  value = whole-job env-steps/s (all GPUs); grad_steps_per_sec beside it; scaling is
weak (per-GPU batch fixed, global batch = N * 4096, gradient all-reduced over RCCL).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 4096
N_STEP = 25
# algorithmic work of the dominant kernel: forward rollout = 26 policy evaluations per start
# state, each 2*(6*256 + 256*256 + 256*2) flop (mean half of the output layer only), + 25 model steps of ~100 flop
FWD_FLOP_PER_STATE = 26 * 2 * (6 * 256 + 256 * 256 + 256 * 2) + 25 * 100
# reverse sweep = 26 input-side backward passes through the same policy (W3^T, W2^T, W1^T) + 25 model adjoints
BWD_FLOP_PER_STATE = 26 * 2 * (2 * 256 + 256 * 256 + 256 * 6) + 25 * 200
FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
# The kernel durations behind `roofline` are measured live, with HIP events on the launch stream, on every PROF_EVERY-th
# launch of the timed region: an event record is a stream packet of its own (~4-5 us between two otherwise back-to-back
# kernels; 8 of them per step with every launch timed = 6 % of a 0.5 ms step - profiles/README.md has the trace).
PROF_EVERY = 16


def build_stack(dev, seed):
    from mpg_amd.buffer import ReplayBuffer
    from mpg_amd.config import default_args
    from mpg_amd.learners import MPGLearner
    from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer
    from mpg_amd.policy import PolicyWithQs
    from mpg_amd.worker import OffPolicyWorker
    args = default_args('MPG-v2', num_agent=B_PER_GPU, batch_size=B_PER_GPU, replay_batch_size=B_PER_GPU,
                        replay_starts=4 * B_PER_GPU, max_buffer_size=500000, seed=seed)
    worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, worker_id=seed, device=dev)
    learner = MPGLearner(PolicyWithQs, args, device=dev)
    rb = ReplayBuffer(args, seed, device=dev)
    opt = SingleProcessOffPolicyOptimizer(worker, learner, rb, None, args, sampling_interval=1)
    worker.policy_with_value.sync_from_rank0()      # replicas start (and, with identical updates, stay) identical
    return args, worker, learner, rb, opt


def cpu_baseline(budget_s=15.0):
    """The oracle ("port": torch-CPU/numpy restatement of the reference, op by op) timed on the host cores on a bounded
    sample of the SAME workload: steps of [4096-agent worker sample + MPG-v2 compute_gradient at B = 4096 + Adam]."""
    from oracle import mpg_oracle as O
    from tests.golden_inputs import mlp_weights_flat
    # torch's intra-op pool: more threads than ~16 only adds overhead on these small tensors (measured: 256 threads
    # on the MI355X host is >100x slower than 8)
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    rng = np.random.Generator(np.random.PCG64(0))
    cfg = O.Cfg()
    names = ['Q1', 'Q2', 'policy']
    flat = {'policy': mlp_weights_flat(rng, 6, 4), 'Q1': mlp_weights_flat(rng, 8, 1), 'Q2': mlp_weights_flat(rng, 8, 1)}
    env = O.PathTrackingEnvOracle(B_PER_GPU)
    env.reset(rng=rng)
    env.obs = env.obs
    n_done, t0 = 0, time.perf_counter()
    while True:
        nets = O.Nets(cfg, flat, target_scale=1.0)
        tr = O.worker_sample(cfg, nets, env, rng, 1)[0]
        batch = [tr[0], tr[1], tr[2], tr[3], tr[4].astype(np.float32)]
        eps = rng.standard_normal((N_STEP, B_PER_GPU)).astype(np.float32)
        grads, _ = O.mpg_compute_gradient(cfg, nets, batch, eps, 100 + n_done, 'MPG-v2')
        n_done += 1
        el = time.perf_counter() - t0
        if el > budget_s or n_done >= 200:
            break
    return {'value': n_done * B_PER_GPU / el, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port',
            'grad_steps_per_sec': n_done / el,
            'sample': '%d steps of [4096-agent worker.sample + MPG-v2 compute_gradient B=4096] in %.1f s, '
                      'torch-CPU oracle, %d threads' % (n_done, el, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    a = ap.parse_args()

    from mpg_amd import dist as D
    import mpg_amd._lib as L
    rank, world, local = D.init_from_env()
    assert world == a.gpus, 'launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (a.gpus, world)
    assert torch.cuda.is_available(), 'bench.py needs a GPU: the product path has no CPU fallback'
    ndev = torch.cuda.device_count()
    if local >= ndev:        # only in MPG_DIST_BACKEND=gloo dry runs with more ranks than GPUs
        assert os.environ.get('MPG_DIST_BACKEND') == 'gloo', 'LOCAL_RANK %d but %d GPUs' % (local, ndev)
        local = local % ndev
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    args, worker, learner, rb, opt = build_stack(dev, seed=rank)

    for _ in range(a.warmup):
        opt.step()
    lib = L.lib()
    D.barrier()
    torch.cuda.synchronize()
    lib.mpg_prof_enable(PROF_EVERY)      # HIP events around every PROF_EVERY-th launch of the timed region (see PROF_EVERY)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        opt.step()
    D.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = D.max_over_ranks(dt)

    def slot(i):
        ms, cnt = ctypes.c_double(0), ctypes.c_int(0)
        L.check(lib.mpg_prof_read(i, ctypes.byref(ms), ctypes.byref(cnt)), 'mpg_prof_read')
        return (ms.value / cnt.value if cnt.value else None), cnt.value
    fwd_ms, fwd_n = slot(0)
    bwd_ms, bwd_n = slot(1)
    env_ms, env_n = slot(2)
    wg_ms, wg_n = slot(5)
    lib.mpg_prof_enable(0)
    n_sampled = (a.steps + PROF_EVERY - 1) // PROF_EVERY
    assert fwd_n == n_sampled and bwd_n == n_sampled, (fwd_n, bwd_n, n_sampled)
    finite = bool(torch.isfinite(worker.policy_with_value.params).all().item())
    assert finite and int(worker.policy_with_value.nonfinite.sum().item()) == 0, 'non-finite parameters after the timed region'

    if rank != 0:
        return
    # HBM bytes per launch from the PMC counters (FETCH_SIZE/WRITE_SIZE, separate rocprofv3 --pmc passes, gfx950
    # correction FETCH x2 for wide coalesced reads) - collected by tools/pmc.sh and committed under profiles/
    traffic = {}
    tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(tpath):
        with open(tpath) as fh:
            traffic = json.load(fh).get('bytes_per_launch', {})

    def roof(kernel, flop, ms, n):
        tf = flop * B_PER_GPU / (ms * 1e-3) / 1e12
        return {'kernel': kernel, 'bound': 'mfma', 'achieved': tf, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': tf / FP32_MFMA_PEAK_TFLOPS, 'traffic': traffic.get(kernel.split('<')[0]), 'avg_ms': ms, 'launches': n,
                'timed_with': 'HIP events on the launch stream around every %d-th launch of the timed region' % PROF_EVERY,
                'algorithmic_flop_per_launch': flop * B_PER_GPU}
    r_fwd = roof('k_rollout_fwd<PathTracking>', FWD_FLOP_PER_STATE, fwd_ms, fwd_n)
    r_bwd = roof('k_rollout_bwd<PathTracking>', BWD_FLOP_PER_STATE, bwd_ms, bwd_n)
    dominant, other = (r_bwd, r_fwd) if bwd_ms >= fwd_ms else (r_fwd, r_bwd)   # the dominant kernel of the step
    out = {
        'metric': 'env-steps/sec + grad-steps/sec, PathTrackingEnv MPG n=25 batch=4096',
        'value': world * B_PER_GPU * a.steps / dt, 'unit': 'env-steps/s',
        'grad_steps_per_sec': a.steps / dt,
        'model_steps_per_sec': world * B_PER_GPU * N_STEP * a.steps / dt,
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * dt / a.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'PathTrackingEnv, MPG-v2 learner, n=25, M=1, 4096 vectorised envs + replay batch 4096 per '
                               'GPU; step = worker.sample(4096 env-steps) + add_batch + replay + compute_gradient + '
                               '(all-reduce) + apply_gradients',
                   'global_batch': world * B_PER_GPU, 'parallelism': 'dp%d' % world,
                   'grad_allreduce_floats': int(learner.flat.numel()), 'native_step_driver': opt._fused is not None},
        'roofline': dominant,
        'roofline_other_rollout_kernel': other,
        'wgrad_kernel': {'kernel': 'k_wgrad_multi', 'avg_ms': wg_ms, 'launches': wg_n},
        'env_step_kernel': {'kernel': 'k_step_store_reset', 'avg_ms': env_ms, 'launches': env_n,
                            'env_steps_per_sec_kernel_only': B_PER_GPU / (env_ms * 1e-3) if env_ms else None,
                            'algorithmic_bytes_per_env_step': 85},
    }
    if world == 1 and not a.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
