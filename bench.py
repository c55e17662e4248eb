#!/usr/bin/env python3
"""bench.py - headline benchmark of the MPG hot path on MI355X.

Workload (BASELINE.json configs[1]): PathTrackingEnv, MPG (learner_version MPG-v2, the reference script's default,
train_script.py:847), n = 25, M = 1, 4096 vectorised envs and a replay batch of 4096 PER GPU.
One "step" = one pass of the hot path over one batch, in SingleProcessOffPolicyOptimizer.step order
(optimizer.py:330-362): worker.sample (policy + N(0, .1) noise -> env.step (20 sub-steps) -> env.reset, 4096 agents)
-> replay_buffer.add_batch -> replay_buffer.replay (4096 rows) -> learner.compute_gradient (clipped double-Q target,
Q1/Q2 loss+grad, 25-step model rollout + mixed policy gradient) -> all-reduce (N > 1) -> clip_by_global_norm ->
worker.apply_gradients (Adam + Polyak).  So every step is 4096 env-steps and 1 gradient step per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5          (the driver's form; defaults: --steps 200 --warmup 20)
    python bench.py --gpus N ...                            (N > 1: this process only spawns the N ranks below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  value = whole-job env-steps/s (all GPUs); grad_steps_per_sec beside it; scaling is
weak (per-GPU batch fixed, global batch = N * 4096, gradient all-reduced over RCCL).

What happens around the timed region (all of it disclosed in the JSON line):
  * `burn_in_steps` (fixed, 200 ~ 0.1 s) untimed steps BEFORE the W warm-up steps: the GPU's clocks and caches take
    ~40-100 steps to settle (kernels are 8-10 % slower until then), which a 5-step warm-up does not cover;
  * Python's cyclic garbage collector is collected + frozen before the burn-in and disabled inside the timed region
    (`gc`): a generation-2 collection of the ~10^6 objects that `import torch` leaves behind takes 40-100 ms, and
    round 1's driver run caught one inside its 20 timed steps (4.62 ms/step against 0.47; tools/stall_scan.py shows it
    at the same step index run after run).  A training loop that cares does the same (`mpg_amd.optimizer.quiesce_gc`);
  * the kernel timer's HIP events all exist before the burn-in (`ops.Profiler`), nothing is created in the loop;
  * after the timed region, a second pass of the same K steps with one HIP event per step gives the per-step
    distribution (`step_ms_median/min/max`); it is NOT part of `value`.
"""
import argparse
import gc
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 4096
N_STEP = 25
BURN_IN_STEPS = int(os.environ.get('MPG_BENCH_BURN_IN', '200'))    # (the multi-rank dry runs of tests/test_dist_gpu.py shorten it)
# Algorithmic work of the two rollout sweeps per start state (n = 25 -> 26 policy evaluations), DESIGN.md §4.
#   bytes (HBM): every evaluation stashes / re-reads the two 256-wide hidden activations in float32 (2 x 1 KiB) plus a
#                32 B (obs | action) record; the reverse sweep also reads 2 x 32 B of critic input gradients and writes the
#                step-0 dz1 / dz2 stash (2 KiB) + dz3 (8 B); the forward sweep reads the 24 B start state and writes 2 x (32 +
#                4) B of critic inputs / reward sums.
#   flops:       2*(6*256 + 256*256 + 256*2) per evaluation forward, 2*(2*256 + 256*256 + 256*6) reverse (+ ~100 / ~200 per
#                model step / adjoint).  The 256 x 256 product runs as THREE f16 MFMAs per fp32-equivalent tile step
#                (split-fp16 engine, csrc/mlp_core.h), i.e. 3x these flops are executed on the f16 pipe.
N_EVAL = N_STEP + 1
FWD_BYTES_PER_STATE = N_EVAL * (2 * 1024 + 32) + 24 + 2 * 36
BWD_BYTES_PER_STATE = N_EVAL * (2 * 1024 + 32) + 2 * 32 + 2 * 1024 + 8
FWD_FLOP_PER_STATE = N_EVAL * 2 * (6 * 256 + 256 * 256 + 256 * 2) + 25 * 100
BWD_FLOP_PER_STATE = N_EVAL * 2 * (2 * 256 + 256 * 256 + 256 * 6) + 25 * 200
HIDDEN_FLOP_PER_STATE = N_EVAL * 2 * 256 * 256          # the part that runs (3x) on the f16 matrix pipe
FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
F16_MFMA_PEAK_TFLOPS = 2500.0          # MI355X_MICROARCH.md: f16 / bf16 dense peak
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
# The kernel durations behind `roofline` are measured live, with HIP events on the launch stream, on every PROF_EVERY-th
# launch of the timed region: an event record is a stream packet of its own (~4-5 us between two otherwise back-to-back
# kernels; 8 of them per step with every launch timed = 6 % of a 0.5 ms step - profiles/README.md has the trace).
PROF_EVERY = 16
N_REGIONS = 5          # timed regions of K steps each; value = the median region
ENV_SAT_AGENTS = 1 << 20      # agents per launch of the stand-alone env-kernel measurement at a saturating size
ENV_FLOP_PER_STEP = 2300.0    # ~2.3 kflop per env-step (20 sub-steps x ~115, transcendentals counted as 1)


def build_stack(dev, seed, always_exchange=False):
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
    opt = SingleProcessOffPolicyOptimizer(worker, learner, rb, None, args, sampling_interval=1, always_exchange=always_exchange)
    worker.policy_with_value.sync_from_rank0()      # replicas start (and, with identical updates, stay) identical
    return args, worker, learner, rb, opt


# ---- CPU baseline: the oracle, the way the reference scales (1 thread per process x P processes) -------------------
CPU_ROWS = 1024            # rows (agents / replay rows) per CPU step: a bounded sample of the 4096-row step (the unit, env-steps/s, scales)
CPU_LEGS = (64, 128, 256)  # processes side by side: per-core regime ... memory-bound regime of a 2-socket host


def _cpu_worker(budget_s, min_timed, seed, q, go, rows=None):
    """One reference-style actor: torch pinned to ONE thread (the reference pins TF the same way, mpg_learner.py:27-28),
    steps of [CPU_ROWS-agent worker.sample + MPG-v2 compute_gradient at B = CPU_ROWS] of the torch-CPU/numpy oracle.  The FIRST step
    (allocator warm-up, first-call costs: BASELINE.md section 2 excludes it too) is not timed; then at least `min_timed` steps and
    as many more as fit into `budget_s`."""
    import numpy as np
    import torch
    torch.set_num_threads(1)
    from oracle import mpg_oracle as O
    from tests.golden_inputs import mlp_weights_flat
    rng = np.random.Generator(np.random.PCG64(seed))
    rows = rows or CPU_ROWS
    cfg = O.Cfg()
    flat = {'policy': mlp_weights_flat(rng, 6, 4), 'Q1': mlp_weights_flat(rng, 8, 1), 'Q2': mlp_weights_flat(rng, 8, 1)}
    env = O.PathTrackingEnvOracle(rows)
    env.reset(rng=rng)

    def step(k):
        nets = O.Nets(cfg, flat, target_scale=1.0)
        tr = O.worker_sample(cfg, nets, env, rng, 1)[0]
        batch = [tr[0], tr[1], tr[2], tr[3], tr[4].astype(np.float32)]
        eps = rng.standard_normal((N_STEP, rows)).astype(np.float32)
        O.mpg_compute_gradient(cfg, nets, batch, eps, 100 + k, 'MPG-v2')
    q.put(('ready', seed))
    go.wait()                   # every process of the leg starts its steps together (imports are not part of the sample)
    step(0)                     # excluded
    n_done, t0 = 0, time.perf_counter()
    while True:
        step(1 + n_done)
        n_done += 1
        el = time.perf_counter() - t0
        if n_done >= min_timed and (el > budget_s or n_done >= 200):
            break
    q.put(('done', n_done, el))


def _cpu_worker_b4096(budget_s, min_timed, seed, q, go):
    """BASELINE.md section 3, run B: ONE reference-style actor at the workload's REAL size - 4096-agent worker.sample + MPG-v2
    compute_gradient at B = 4096 per step, one thread (mpg_learner.py:25-28)"""
    _cpu_worker(budget_s, min_timed, seed, q, go, rows=4096)


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _host_limits(max_procs):
    try:
        host = len(os.sched_getaffinity(0))
    except AttributeError:
        host = os.cpu_count() or 1
    try:
        import psutil
        mem_cap = max(1, int(psutil.virtual_memory().available * 0.5 / 1.5e9))
    except Exception:
        mem_cap = 8
    return host, mem_cap, max(1, min(host, max_procs, mem_cap))


def _run_leg(target, args_of, procs, timeout_s):
    """`procs` spawned single-threaded processes side by side; returns their ('done', units, seconds) results and the wall time"""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')          # fresh interpreters: nothing of this process' GPU state is inherited
    q, go = ctx.Queue(), ctx.Event()
    ps = [ctx.Process(target=target, args=args_of(i) + (q, go)) for i in range(procs)]
    t0 = time.perf_counter()
    for p in ps:
        p.start()
    ready, res = 0, []
    try:
        while ready < procs:
            q.get(timeout=300)
            ready += 1
        go.set()
        for _ in ps:
            res.append(q.get(timeout=timeout_s)[1:])
    except Exception:              # noqa: BLE001 - a process that died or stalled: report what arrived
        go.set()
    for p in ps:
        p.join(timeout=10)
        if p.is_alive():
            p.kill()
    return res, time.perf_counter() - t0


def cpu_baseline(budget_s=8.0, min_timed=3, max_procs=256):
    """`kind: port` - the oracle timed on the host cores on a bounded sample of the SAME workload, shaped like the
    reference's own scaling: P single-threaded processes side by side, for P = 64 / 128 / 256 (capped by
    the host's cores and by what the free memory allows at ~1.5 GB per process: `import torch` alone is 0.6 GB of private
    pages), so that the per-core regime (few processes) and the memory-bound regime (every hardware thread busy) are both in
    the line.  Every process: one untimed step, then >= 3 timed steps of CPU_ROWS rows.  `value` = the best leg."""
    host, mem_cap, cap = _host_limits(max_procs)
    # one thread per process means ONE: without these every child would start an OpenMP/MKL pool as wide as the host
    # (256 threads each on the MI355X box) before torch.set_num_threads(1) is reached
    for k in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'NUMEXPR_NUM_THREADS'):
        os.environ[k] = '1'
    os.environ['HIP_VISIBLE_DEVICES'] = ''      # the CPU actors never touch the GPU
    legs = []
    for want in sorted(set(min(p, cap) for p in CPU_LEGS)):
        res, wall = _run_leg(_cpu_worker, lambda i: (budget_s, min_timed, 1000 + i), want, budget_s * 20 + 300)
        if not res:
            continue
        sps = sum(n / el for n, el in res)
        legs.append({'processes': len(res), 'env_steps_per_sec': sps * CPU_ROWS, 'steps_per_sec_per_process': sps / len(res),
                     'seconds_per_step': len(res) / sps, 'timed_steps': sum(n for n, _ in res), 'wall_s': round(wall, 1)})
    best = max(legs, key=lambda g: g['env_steps_per_sec'])
    # leg B: the SAME learner at the workload's real size - one single-threaded process, 4096 rows per step,
    # one untimed step and then >= 3 timed ones - beside the scaled 1024-row legs
    full = {}
    res, wall = _run_leg(_cpu_worker_b4096, lambda i: (budget_s, min_timed, 2000 + i), 1, 900)
    if res:
        n, el = res[0]
        full = {'b4096_processes': 1, 'b4096_rows_per_cpu_step': 4096, 'b4096_timed_steps': n, 'b4096_seconds_per_step': el / n,
                'b4096_env_steps_per_sec': 4096 * n / el, 'b4096_grad_steps_per_sec': n / el, 'b4096_wall_s': round(wall, 1)}
    out = {'value': best['env_steps_per_sec'], 'unit': 'env-steps/s', 'cores': best['processes'], 'kind': 'port',
           'grad_steps_per_sec': best['env_steps_per_sec'] / B_PER_GPU, 'host_cores': host, 'cpu_model': _cpu_model(),
           'rows_per_cpu_step': CPU_ROWS, 'legs': legs,
           'sample': 'P single-threaded processes side by side, P = %s (host cores %d, free-memory cap %d); each process: one '
                     'untimed step, then >= %d timed steps (%.0f s budget) of [%d-agent worker.sample + MPG-v2 compute_gradient '
                     'B=%d] of the torch-CPU oracle - a quarter of the 4096-row step per CPU step, env-steps/s scales with the rows; '
                     'value = the best leg (%d processes); grad_steps_per_sec = value / 4096; b4096_*: ONE process at the real '
                     '4096 rows per step (BASELINE.md section 3 run B)'
                     % ('/'.join(str(g['processes']) for g in legs), host, mem_cap, min_timed, budget_s, CPU_ROWS, CPU_ROWS, best['processes'])}
    # flat keys (the driver's record keeps scalars of this object): per-leg rates with the rows each CPU step held
    for g in legs:
        out['p%d_env_steps_per_sec' % g['processes']] = g['env_steps_per_sec']
        out['p%d_rows_per_cpu_step' % g['processes']] = CPU_ROWS
    out.update(full)
    return out


def _flush_c_stdio():
    """RCCL prints a version banner through C stdio, which a pipe buffers until exit - i.e. BEHIND the JSON line.  Flushing it first
    keeps the JSON line the last line of stdout."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                # noqa: BLE001
        pass


def _under_profiler():
    """rocprofv3 (or any rocprofiler-sdk tool) preloads its library into this process AND into every child: the children's kernels
    would be averaged into the profile of this run under the same names"""
    if 'rocprof' in os.environ.get('LD_PRELOAD', '').lower():
        return True
    return any(k.startswith(('ROCPROF', 'ROCP_TOOL')) for k in os.environ)


def _env_sat(sat):
    out = {'agents_per_launch': ENV_SAT_AGENTS, 'algorithmic_bytes_per_env_step': 85, 'algorithmic_flop_per_env_step': ENV_FLOP_PER_STEP,
           'timed_with': 'one HIP event pair around 10 back-to-back launches after the timed region (3 untimed first)'}
    for name, ms in sat.items():
        rate = ENV_SAT_AGENTS / (ms * 1e-3)
        f_hbm, f_valu = 85 * rate / 1e9 / HBM_PEAK_GBS, ENV_FLOP_PER_STEP * rate / 1e12 / FP32_MFMA_PEAK_TFLOPS
        out[name] = {'kernel': 'k_step (mpg_env_step)' if name == 'step_only' else 'k_step_store_reset (mpg_env_step_store_reset)',
                     'avg_ms': ms, 'env_steps_per_sec': rate, 'achieved_GBs': 85 * rate / 1e9, 'frac_hbm': f_hbm,
                     'achieved_valu_tflops': ENV_FLOP_PER_STEP * rate / 1e12, 'peak_valu_tflops': FP32_MFMA_PEAK_TFLOPS, 'frac_valu': f_valu,
                     'binding': 'valu' if f_valu >= f_hbm else 'hbm'}
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(a):
    """`python bench.py --gpus N` from a plain shell: this process touches no GPU and starts the N ranks as children
    (torch.distributed.run, one process per GPU), then leaves with their exit code."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus), '--master-addr',
           '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__), '--gpus', str(a.gpus), '--steps',
           str(a.steps), '--warmup', str(a.warmup), '--config', a.config, '--rows-per-gpu', str(a.rows_per_gpu)] + (['--no-cpu-baseline'] if a.no_cpu_baseline else [])
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '1')
    return subprocess.call(cmd, env=env)


def _device_info():
    """Name, CU count and clocks of the GPU the line was measured on (boxes of the pool differ: see DESIGN.md section 6)."""
    import torch
    p = torch.cuda.get_device_properties(torch.cuda.current_device())
    info = {'name': p.name, 'compute_units': p.multi_processor_count, 'hbm_gib': round(p.total_memory / 2 ** 30, 1)}
    for k in ('clock_rate', 'memory_clock_rate'):
        if hasattr(p, k):
            info[k + '_khz'] = getattr(p, k)
    return info


# ---- side configurations C3 / C4 (BASELINE.json configs[2], configs[3]) ---------------------------------------------------
# Algorithmic work per unit, SURVEY.md section 8d: NADP step 14.8 Mflop per start state (pi 4->256->256->2, Q 5->256->256->1,
# 55 MLP evaluations forward + as many reverse), TD3 step 2.17 Mflop per replay row, PER 212 B per sampled index at
# B = 65 536 / N = 2^19.  The hidden-layer products run 3x on the f16 matrix pipe (split operands), so the matrix view prices
# 3x the hidden-layer part against the f16 dense peak - the pipe actually used.
SIDE = {
    'c3': dict(B=8192, alg='NADP', flop_per_row=14.8e6, hidden_flop_per_row=110 * 2 * 256 * 256,
               metric='grad-steps/sec, InvertedPendulumConti model NADP n=25 batch=8192',
               workload='InvertedPendulumConti (inverted_pendulum_model.py), NADP learner, n=25, replay batch 8192 per GPU; step = '
                        'replay (uniform draw + gather) + compute_gradient (Q-target rollout + Q loss/grad + policy rollout with all-step parameter gradients) + '
                        '(all-reduce) + apply_gradients'),
    'c4': dict(B=65536, alg='TD3', flop_per_row=2.17e6, hidden_flop_per_row=16 * 2 * 256 * 256,
               metric='replay-rows/sec + grad-steps/sec, PathTrackingEnv TD3 + prioritized replay batch=65536',
               workload='PathTrackingEnv, TD3 learner, prioritized replay (segment trees, capacity 2^19, 500k transitions), replay '
                        'batch 65536 per GPU; step = replay (proportional sampling + gather) + compute_gradient + update_priorities + '
                        '(all-reduce) + apply_gradients'),
}


def _cpu_side_worker(budget_s, seed, config, q, go):
    """one single-threaded actor running the oracle's gradient step of the side configuration on a REDUCED batch (the unit is
    rows / s, so the sample scales): NADP 512 rows, TD3 4096 rows; first step untimed"""
    import numpy as np
    import torch
    torch.set_num_threads(1)
    from oracle import mpg_oracle as O
    from tests.golden_inputs import mlp_weights_flat, reset_law_obs
    rng = np.random.Generator(np.random.PCG64(seed))
    if config == 'c3':
        rows = 512
        cfg = O.Cfg(env='InvertedPendulumConti-v0', select=[25], delay_update=1)
        flat = {'policy': mlp_weights_flat(rng, 4, 2), 'Q1': mlp_weights_flat(rng, 5, 1)}
        obs = (rng.standard_normal((rows, 4)) * np.array([0.5, 0.1, 0.5, 0.5])).astype(np.float32)
        act = rng.uniform(-3, 3, (rows, 1)).astype(np.float32)
    else:
        rows = 4096
        cfg = O.Cfg()
        flat = {'policy': mlp_weights_flat(rng, 6, 4), 'Q1': mlp_weights_flat(rng, 8, 1), 'Q2': mlp_weights_flat(rng, 8, 1)}
        batch = [reset_law_obs(rng, rows), rng.uniform(-1, 1, (rows, 2)).astype(np.float32), rng.uniform(-30, 0, rows).astype(np.float32),
                 reset_law_obs(rng, rows), np.ones(rows, np.float32)]

    def step():
        nets = O.Nets(cfg, flat, target_scale=1.0)
        if config == 'c3':
            O.nadp_compute_gradient(cfg, nets, [obs, act], rng.standard_normal((25, rows)).astype(np.float32),
                                    rng.standard_normal((25, rows)).astype(np.float32))
        else:
            O.td3_compute_gradient(cfg, nets, batch, rng.standard_normal((rows, 2)).astype(np.float32))
    q.put(('ready', seed))
    go.wait()
    step()
    n_done, t0 = 0, time.perf_counter()
    while True:
        step()
        n_done += 1
        el = time.perf_counter() - t0
        if n_done >= 3 and (el > budget_s or n_done >= 500):
            break
    q.put(('done', n_done * rows, el))


def cpu_side_baseline(config, budget_s=8.0, max_procs=256):
    host, mem_cap, procs = _host_limits(max_procs)
    for k in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'NUMEXPR_NUM_THREADS'):
        os.environ[k] = '1'
    os.environ['HIP_VISIBLE_DEVICES'] = ''
    res, _ = _run_leg(_cpu_side_worker, lambda i: (budget_s, 2000 + i, config), procs, budget_s * 20 + 300)
    rows_per_s = sum(n / el for n, el in res)
    return {'value': rows_per_s, 'unit': 'rows/s', 'cores': len(res), 'kind': 'port', 'host_cores': host, 'cpu_model': _cpu_model(),
            'sample': '%d single-threaded processes side by side, each one untimed step and then >= 3 timed steps (%.0f s budget) of the '
                      'oracle\'s %s gradient step on %d-row batches (rows/s scales with the batch); process count = min(host cores %d, '
                      '%d, free-memory cap %d)' % (len(res), budget_s, SIDE[config]['alg'], 512 if config == 'c3' else 4096, host, max_procs, mem_cap)}


def side_config(a):
    import numpy as np
    import torch
    from mpg_amd import dist as D
    from mpg_amd import ops
    from mpg_amd.config import default_args
    from mpg_amd.optimizer import quiesce_gc
    from mpg_amd.policy import PolicyWithQs
    rank, world, local = D.init_from_env()
    assert world == a.gpus and torch.cuda.is_available()
    ndev = torch.cuda.device_count()
    local = local % ndev
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    c = SIDE[a.config]
    B = c['B']
    rng = np.random.Generator(np.random.PCG64(rank))
    from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer
    from mpg_amd.worker import OffPolicyWorker
    # the stock stack in SingleProcessOffPolicyOptimizer's order; `--side-driver native` (default): enqueued by the native step driver
    # (mpg_step_begin / _end, learner_version 3 / 4), `method`: method by method from Python (rounds 2 - 4).  The worker's sampling
    # (every 10th iteration of 512 env steps in the reference, optimizer.py:331) is NOT part of these side lines' step: the interval is
    # set beyond the run, the replay is filled up front.
    never = 1 << 30
    if a.config == 'c3':
        from mpg_amd.buffer import ReplayBuffer
        from mpg_amd.learners import NADPLearner
        args = default_args('NADP', replay_batch_size=B, num_agent=512, batch_size=512, replay_starts=2 * B, seed=rank)
        worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, rank, device=dev)
        learner = NADPLearner(PolicyWithQs, args, device=dev)
        rb = ReplayBuffer(args, rank, device=dev)
    else:
        from mpg_amd.buffer import PrioritizedReplayBuffer
        from mpg_amd.learners import TD3Learner
        from tests.golden_inputs import reset_law_obs
        N = 500000
        args = default_args('TD3', replay_batch_size=B, buffer_type='priority', replay_starts=N, num_agent=512, batch_size=512, seed=rank)
        worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, rank, device=dev)
        learner = TD3Learner(PolicyWithQs, args, device=dev)
        rb = PrioritizedReplayBuffer(args, rank, device=dev)
        for _ in range(N // 50000):
            rb.add_batch((torch.as_tensor(reset_law_obs(rng, 50000)).to(dev), torch.as_tensor(rng.uniform(-1, 1, (50000, 2)), dtype=torch.float32).to(dev),
                          torch.as_tensor(rng.standard_normal(50000), dtype=torch.float32).to(dev),
                          torch.as_tensor(reset_law_obs(rng, 50000)).to(dev), torch.ones(50000, dtype=torch.uint8, device=dev)))
    opt = SingleProcessOffPolicyOptimizer(worker, learner, rb, None, args, sampling_interval=never, fused=a.side_driver == 'native')
    opt.iteration = 1                       # (iteration 0 would sample: 0 % interval == 0)
    assert (opt._fused is not None) == (a.side_driver == 'native')
    pw = worker.policy_with_value
    pw.sync_from_rank0()
    prof = ops.Profiler(max_samples=4096)
    opt.set_profiler(prof)
    step = opt.step
    quiesce_gc()
    for _ in range(max(a.warmup, 5)):
        step()
    gc.disable()
    regions = []
    for _ in range(N_REGIONS):              # like the main line: value from the median of N_REGIONS regions of K steps
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        D.barrier()
        torch.cuda.synchronize()
        regions.append(D.max_over_ranks(time.perf_counter() - t0))
    gc.enable()
    dt = sorted(regions)[len(regions) // 2]
    # kernel groups: a second, short pass with EVERY launch of the library's timer slots bracketed by HIP events (the records
    # cost ~4-5 us each, which is why this pass is not the timed one)
    nprof = min(a.steps, 10)
    prof.start(1)
    for _ in range(nprof):
        step()
    torch.cuda.synchronize()
    slots = {}
    for k in range(8):
        ms, n = prof.read(k)
        if n:
            slots[ctypes_name(k)] = {'avg_ms': ms, 'launches_per_step': n / nprof, 'ms_per_step': ms * n / nprof}
    prof.stop()
    t_per = None
    if a.config == 'c4':    # the sampling + gather alone, timed with device events on the launch stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            rb.replay()
        e1.record()
        torch.cuda.synchronize()
        t_per = e0.elapsed_time(e1) / 20
    assert bool(torch.isfinite(pw.params).all().item()) and int(pw.nonfinite.sum().item()) == 0
    pw.check_status()
    if rank != 0:
        return
    ms = 1e3 * dt / a.steps
    dom = max(slots, key=lambda k: slots[k]['ms_per_step']) if slots else None
    executed = (c['flop_per_row'] + 2 * c['hidden_flop_per_row']) * B / (ms * 1e-3) / 1e12
    algorithmic = c['flop_per_row'] * B / (ms * 1e-3) / 1e12
    out = {
        'metric': c['metric'], 'value': world * B * a.steps / dt, 'unit': 'rows/s', 'grad_steps_per_sec': a.steps / dt,
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak',
        'timed_regions': N_REGIONS, 'region_ms_per_step': [1e3 * r / a.steps for r in regions],
        'vs_baseline': None, 'dtype': 'f32-via-split-f16', 'data': 'synthetic', 'schema': 4,
        'config': {'workload': c['workload'], 'global_batch': world * B, 'parallelism': 'dp%d' % world, 'dist_backend': D.backend(),
                   'driver': 'native step driver (mpg_step_begin / _end)' if a.side_driver == 'native' else 'method by method from Python'},
        'device': _device_info(),
        # whole-step matrix view (the step is a chain of weight-stationary launches, all on the f16 pipe).  `achieved` / `frac`
        # count ALGORITHMIC flop (the 3x of the split-fp16 emulation is overhead, not work); the executed view (the
        # hidden-layer part counted 3x) stays under frac_f16_mfma
        'roofline': {'bound': 'mfma', 'kernel': 'whole gradient step (%s dominates: %.3f ms of it)' % (dom, slots[dom]['ms_per_step']) if dom else 'whole gradient step',
                     'achieved': algorithmic, 'peak': F16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': algorithmic / F16_MFMA_PEAK_TFLOPS,
                     'frac_f16_mfma': executed / F16_MFMA_PEAK_TFLOPS, 'executed_f16_tflops': executed, 'traffic': None,
                     'algorithmic_tflops': algorithmic,
                     'timed_with': 'wall clock over the timed region (kernel groups by HIP events on the launch stream below)'},
        'kernel_groups_ms_per_step': slots,
    }
    if t_per is not None:
        out['per'] = {'sample_gather_ms': t_per, 'algorithmic_bytes_per_index': 212, 'achieved_GBs': 212 * B / (t_per * 1e-3) / 1e9,
                      'peak_GBs': HBM_PEAK_GBS, 'frac_hbm': 212 * B / (t_per * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if world == 1 and not a.no_cpu_baseline:
        out['cpu_baseline'] = cpu_side_baseline(a.config)
    print(json.dumps(out), flush=True)


def ctypes_name(slot):
    from mpg_amd import _lib as L
    return L.lib().mpg_prof_slot_name(slot).decode()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--side-driver', default='native', choices=['native', 'method'],
                    help='--config c3 / c4: the native step driver (default) or the method-by-method path of rounds 2 - 4')
    ap.add_argument('--always-exchange', action='store_true',
                    help='diagnostic, one GPU: run the exchange path in a ONE-rank RCCL group (the collective between mpg_step_begin and '
                         'mpg_step_end, the clip partials from the exchanged buffer; with MPG_OVERLAP_EXCHANGE=1 also the critics\' early '
                         'gradient + second stream) - what the multi-GPU step costs on one GPU beyond the bytes on the wire')
    ap.add_argument('--no-side-configs', action='store_true', help='do not run --config c3 / c4 as child processes after the timed region')
    ap.add_argument('--config', default='c2', choices=['c2', 'c3', 'c4'],
                    help='c2 (default): the BASELINE metric - PathTracking MPG n=25 batch 4096; c3: NADP on the pendulum model, batch '
                         '8192; c4: TD3 + prioritized replay, batch 65536 (BASELINE.json configs[2], [3]: side lines, same JSON shape)')
    ap.add_argument('--rows-per-gpu', type=int, default=4096,
                    help='c2 only: agents AND replay rows per GPU (default 4096 = the BASELINE metric; 32768 = config 5\'s GLOBAL batch on one GPU, '
                         'the N = 1 anchor of its strong-scaling reading - the default run starts that form as a child: side_configs.c5_on_1gpu)')
    ap.add_argument('--engine', default='split', choices=['split', 'f32'],
                    help='split (default): the product; f32: the same step on libmpg_hip_f32.so, the exact-fp32 engine (the main run '
                         'starts this form itself as a child process and reports it as exact_fp32_ms_per_step)')
    a = ap.parse_args()
    global B_PER_GPU
    assert a.rows_per_gpu % 16 == 0 and a.rows_per_gpu >= 4096, '--rows-per-gpu: a multiple of 16, at least 4096'
    B_PER_GPU = a.rows_per_gpu
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(a))
    from mpg_amd import _lib as LIBSEL
    LIBSEL.select_engine(a.engine)
    if a.config != 'c2':
        return side_config(a)

    import torch
    from mpg_amd import dist as D
    from mpg_amd import ops
    from mpg_amd.optimizer import quiesce_gc
    if a.always_exchange and a.gpus == 1 and 'WORLD_SIZE' not in os.environ:
        os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    # (--always-exchange on one GPU: a ONE-rank RCCL group, or - MPG_DIST_BACKEND=oneshot - the one-shot IPC exchange with itself)
    rank, world, local = D.init_from_env(backend=(os.environ.get('MPG_DIST_BACKEND') or 'nccl') if a.always_exchange and a.gpus == 1 else None)
    assert world == a.gpus, '--gpus %d but WORLD_SIZE=%d' % (a.gpus, world)
    assert torch.cuda.is_available(), 'bench.py needs a GPU: the product path has no CPU fallback'
    ndev = torch.cuda.device_count()
    if local >= ndev:        # only in MPG_DIST_BACKEND=gloo / oneshot dry runs with more ranks than GPUs
        assert os.environ.get('MPG_DIST_BACKEND') in ('gloo', 'oneshot'), 'LOCAL_RANK %d but %d GPUs' % (local, ndev)
        local = local % ndev
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    args, worker, learner, rb, opt = build_stack(dev, seed=rank, always_exchange=a.always_exchange)
    n_samples = (N_REGIONS * max(a.steps, 1) + PROF_EVERY - 1) // PROF_EVERY + 1
    prof = ops.Profiler(max_samples=n_samples)      # every HIP event exists from here on
    opt.set_profiler(prof)
    quiesce_gc()                                     # gc.collect() + gc.freeze(): see the module docstring
    prof.start(PROF_EVERY)
    for _ in range(BURN_IN_STEPS):
        opt.step()
    for _ in range(a.warmup):
        opt.step()
    prof.start(PROF_EVERY)                           # clears the samples; no event is created
    gc.disable()
    # N_REGIONS consecutive timed regions of EXACTLY K steps each, every one bracketed by barrier + synchronize on both sides and
    # reduced with MAX over the ranks; `value` comes from the MEDIAN region (one 4.7 ms region is at the mercy of
    # a single host stall - profiles/r04_final_bench_driver_form_outlier.json - so the line lists all of them)
    regions = []
    for _ in range(N_REGIONS):
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            opt.step()
        D.barrier()
        torch.cuda.synchronize()
        regions.append(D.max_over_ranks(time.perf_counter() - t0))
    gc.enable()
    dt = sorted(regions)[len(regions) // 2]

    fwd_ms, fwd_n = prof.read(0)
    bwd_ms, bwd_n = prof.read(1)
    env_ms, env_n = prof.read(2)
    pol_ms, pol_n = prof.read(3)
    wg_ms, wg_n = prof.read(5)
    tgt_ms, tgt_n = prof.read(6)
    crit_ms, crit_n = prof.read(7)
    xch_ms, xch_n = prof.read(8)                     # the gradient exchange (None on one GPU: there is none)
    adam_ms, adam_n = prof.read(9)
    prof.stop()
    n_sampled = (N_REGIONS * a.steps + PROF_EVERY - 1) // PROF_EVERY
    assert fwd_n == n_sampled and bwd_n == n_sampled, (fwd_n, bwd_n, n_sampled)
    # per-step distribution: a second pass of the same K steps, one HIP event per step (not part of `value`)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    gc.disable()
    ev[0].record()
    for i in range(a.steps):
        opt.step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    gc.enable()
    per_step = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps))
    # env `step` alone (SURVEY section 8d asks for BOTH env rates: step only - worker.py:108 - and step + store + reset, the fused
    # launch the training step uses - worker.py:108-112): 64 back-to-back launches of mpg_env_step on a scratch copy of the
    # worker's state, one HIP event pair around the batch on the launch stream (not part of `value`)
    from mpg_amd import _lib as L
    e_state, e_act = worker.env._state.clone(), torch.zeros(B_PER_GPU, 2, device=dev)
    e_obs, e_rew = torch.empty(B_PER_GPU, 6, device=dev), torch.empty(B_PER_GPU, device=dev)
    e_done, e_di = torch.empty(B_PER_GPU, dtype=torch.uint8, device=dev), torch.empty(B_PER_GPU, dtype=torch.uint8, device=dev)
    es = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def env_step_only(k):
        for _ in range(k):
            L.call('mpg_env_step', L.c_int(0), L.c_int(B_PER_GPU), L.c_int(6), L.ptr(e_state), L.ptr(e_act), L.ptr(e_obs), L.ptr(e_rew),
                   L.ptr(e_done), L.ptr(e_di), L.stream())
    env_step_only(8)
    torch.cuda.synchronize()
    torch.cuda._sleep(40000000)                      # ~20 ms of spin on the stream: the 64 launches queue up behind it and then
    es[0].record()                                   # run back to back (enqueued one by one from Python they would be host-paced)
    env_step_only(64)
    es[1].record()
    torch.cuda.synchronize()
    step_only_ms = es[0].elapsed_time(es[1]) / 64
    # ... and env step + store + reset alone (mpg_env_step_store_reset on scratch copies of the worker's state and a scratch ring): in the
    # training step it is part of the fused worker launch (mpg_worker_step: policy pass + env), whose time is reported separately
    r_cap = 4 * B_PER_GPU
    r_obs, r_obs2 = torch.empty(r_cap, 6, device=dev), torch.empty(r_cap, 6, device=dev)
    r_act, r_rew, r_done = torch.empty(r_cap, 2, device=dev), torch.empty(r_cap, device=dev), torch.empty(r_cap, dtype=torch.uint8, device=dev)

    def env_step_store_reset(k):
        for j in range(k):
            L.call('mpg_env_step_store_reset', L.c_int(0), L.c_int(B_PER_GPU), L.c_int(6), L.ptr(e_state), L.ptr(e_act), L.c_int(r_cap),
                   L.c_int((j % 4) * B_PER_GPU), L.ptr(r_obs), L.ptr(r_act), L.ptr(r_rew), L.ptr(r_obs2), L.ptr(r_done), L.c_u64(7), L.c_u64(j),
                   L.ptr(e_obs), L.ptr(e_done), L.stream())
    env_step_store_reset(8)
    torch.cuda.synchronize()
    torch.cuda._sleep(40000000)
    es[0].record()
    env_step_store_reset(64)
    es[1].record()
    torch.cuda.synchronize()
    step_store_reset_ms = es[0].elapsed_time(es[1]) / 64
    # ... and the env kernel at a SATURATING size (SURVEY section 8d K1 prices it as a scan: 85 B per env-step against the HBM roof,
    # with the VALU fraction beside it - 4096 agents are 64 waves on 256 CUs, which measures launch latency and the 20-sub-step
    # dependency chain, not the kernel): ENV_SAT_AGENTS agents per launch, reset law for the states, U(-1, 1) actions
    ns = ENV_SAT_AGENTS
    from mpg_amd.envs import PathTrackingEnv
    big = PathTrackingEnv(num_agent=ns, device=dev, seed=11)
    big.reset()
    b_act = torch.rand(ns, 2, device=dev) * 2 - 1
    b_obs, b_rew = torch.empty(ns, 6, device=dev), torch.empty(ns, device=dev)
    b_done, b_di = torch.empty(ns, dtype=torch.uint8, device=dev), torch.empty(ns, dtype=torch.uint8, device=dev)
    b_robs, b_robs2 = torch.empty(ns, 6, device=dev), torch.empty(ns, 6, device=dev)
    b_ract, b_rrew, b_rdone = torch.empty(ns, 2, device=dev), torch.empty(ns, device=dev), torch.empty(ns, dtype=torch.uint8, device=dev)

    def big_step(k):
        for _ in range(k):
            L.call('mpg_env_step', L.c_int(0), L.c_int(ns), L.c_int(6), L.ptr(big._state), L.ptr(b_act), L.ptr(b_obs), L.ptr(b_rew),
                   L.ptr(b_done), L.ptr(b_di), L.stream())

    def big_step_store_reset(k):
        for j in range(k):
            L.call('mpg_env_step_store_reset', L.c_int(0), L.c_int(ns), L.c_int(6), L.ptr(big._state), L.ptr(b_act), L.c_int(ns), L.c_int(0),
                   L.ptr(b_robs), L.ptr(b_ract), L.ptr(b_rrew), L.ptr(b_robs2), L.ptr(b_rdone), L.c_u64(7), L.c_u64(j), L.ptr(b_obs),
                   L.ptr(b_done), L.stream())
    sat = {}
    for name, fn in (('step_only', big_step), ('step_store_reset', big_step_store_reset)):
        fn(3)
        torch.cuda.synchronize()
        es[0].record()
        fn(10)
        es[1].record()
        torch.cuda.synchronize()
        sat[name] = es[0].elapsed_time(es[1]) / 10
    del big, b_act, b_obs, b_rew, b_done, b_di, b_robs, b_robs2, b_ract, b_rrew, b_rdone
    finite = bool(torch.isfinite(worker.policy_with_value.params).all().item())
    assert finite and int(worker.policy_with_value.nonfinite.sum().item()) == 0, 'non-finite parameters after the timed region'
    worker.policy_with_value.check_status()          # raises if the split-fp16 engine left its numerical envelope anywhere
    # (checked HERE, before the side measurement below: an abandoned collective may leave the stream blocked for good)
    # N > 1: the exchange on its own in every form this process group can run (RCCL when the group is an nccl group; the one-shot and
    # the two-shot IPC forms of mpg_amd/dist.py), 50 exchanges of a scratch buffer of the gradient's length each, HIP events on the
    # launch stream of rank 0.  A form that cannot be set up (no peer access, IPC refused) is reported as its error text on EVERY rank
    # (the constructor fails collectively) - the line still prints; nothing here is part of `value`.
    exchange_forms, side_hung = None, False
    if world > 1:
        import threading
        import torch.distributed as tdist
        exchange_forms = {}
        scratch_buf = torch.zeros_like(learner.flat)

        def time_form(fn):
            for _ in range(5):
                fn(scratch_buf)
            D.barrier()
            torch.cuda.synchronize()
            es[0].record()
            for _ in range(50):
                fn(scratch_buf)
            es[1].record()
            torch.cuda.synchronize()
            return es[0].elapsed_time(es[1]) / 50

        def guarded(fn, seconds=90.0):
            """a SIDE measurement must never take the line down: it runs in a daemon thread with a deadline (a collective that a
            peer never joins cannot be cancelled, but it can be abandoned - the process then leaves through os._exit below)"""
            box = {}

            def run():
                try:
                    torch.cuda.set_device(dev)
                    box['v'] = fn()
                except Exception as e:           # noqa: BLE001
                    box['e'] = e
            th = threading.Thread(target=run, daemon=True)
            th.start()
            th.join(seconds)
            if th.is_alive():
                return 'timeout', None
            return ('error', box['e']) if 'e' in box else ('ok', box['v'])
        forms = [('nccl', lambda: time_form(lambda b: tdist.all_reduce(b, op=tdist.ReduceOp.SUM)))] if tdist.get_backend() == 'nccl' else []
        for form in ('oneshot', 'twoshot'):
            forms.append((form, lambda form=form: time_form(D.OneShotAllReduce(scratch_buf.numel(), dev, mode=form).all_reduce_sum_)))
        for name, fn in forms:
            status, v = guarded(fn)
            if status == 'ok':
                exchange_forms[name] = v
            elif status == 'error':
                exchange_forms[name] = 'unavailable: %s' % (str(v)[:300],)
            else:
                exchange_forms[name] = 'abandoned after 90 s (a rank did not join); later forms not attempted'
                side_hung = True
                break
        if side_hung and rank != 0:
            os._exit(0)                          # this rank's part of the line (the timed regions) is done
    if rank != 0:
        return
    # HBM bytes per launch from the PMC counters (FETCH_SIZE/WRITE_SIZE, separate rocprofv3 --pmc passes, gfx950
    # correction FETCH x2 for wide coalesced reads) - collected by tools/pmc.sh and committed under profiles/
    traffic, traffic_from = {}, None
    tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(tpath):
        with open(tpath) as fh:
            tj = json.load(fh)
        traffic = tj.get('bytes_per_launch', {})
        traffic_from = 'profiles/pmc_traffic.json (%s) - a committed PMC profile, NOT a measurement of this run' % tj.get('from', 'tools/pmc.sh')
    # the same command on the exact-fp32 engine (libmpg_hip_f32.so = -DMPG_F32_MFMA: v_mfma_f32_16x16x4_f32, no fp16 operand
    # anywhere), measured NOW: a child process (a fresh interpreter that selects the other shared object before its first GPU
    # call; this process only waits) runs the same K / W after this run's timed region
    exact = {'ms_per_step': None, 'from': None}
    profiled = _under_profiler()
    children = a.engine == 'split' and world == 1 and not os.environ.get('MPG_BENCH_NO_F32') and not profiled and a.rows_per_gpu == 4096

    def child(extra, timeout=900):
        cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', str(a.steps), '--warmup', str(a.warmup), '--no-cpu-baseline'] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        if r.returncode == 0 and line:
            return json.loads(line[-1]), ' '.join(cmd[1:])
        raise RuntimeError('rc %d: %s' % (r.returncode, r.stderr[-300:]))
    if children:
        try:
            d32, how = child(['--engine', 'f32', '--no-side-configs'])
            exact = {'ms_per_step': d32['ms_per_step'], 'from': 'this run: child process `%s` after the timed region' % how,
                     'rollout_kernels_ms': [d32['roofline']['avg_ms'], d32['roofline_other_rollout_kernel']['avg_ms']]}
        except Exception as e:                       # noqa: BLE001 - a side measurement must not take the line down
            exact['from'] = 'child failed: %r' % (e,)
    elif profiled:
        exact['from'] = 'skipped: this process runs under a profiler (the child would inherit its preload and be counted into its output)'
    # BASELINE.json configs[2] / [3] (C3: NADP on the pendulum model, B = 8192; C4: TD3 + prioritized replay, B = 65 536) as
    # child processes after the timed region, like the exact-fp32 engine: parity-test configurations, NOT the bench line - here so
    # that the driver's record carries them
    side = {}
    if children and not a.no_side_configs:
        for cname in ('c3', 'c4'):
            try:
                d, how = child(['--config', cname])
                side[cname] = {k: d.get(k) for k in ('metric', 'value', 'unit', 'ms_per_step', 'grad_steps_per_sec', 'steps', 'timed_regions',
                                                     'region_ms_per_step', 'roofline', 'per', 'kernel_groups_ms_per_step')}
                side[cname]['from'] = 'child process `%s` after the timed region' % how
            except Exception as e:                   # noqa: BLE001
                side[cname] = {'error': repr(e)}
        # BASELINE.json configs[4] is 8 x 4096 rows over 8 GPUs; its GLOBAL batch (32 768 agents + replay batch 32 768) on ONE GPU is the
        # numerator of its strong-scaling speed-up and the throughput regime of this engine (8 row groups per CU in the sweeps instead of 1)
        try:
            d, how = child(['--rows-per-gpu', '32768', '--no-side-configs'])
            side['c5_on_1gpu'] = {k: d.get(k) for k in ('metric', 'value', 'unit', 'ms_per_step', 'grad_steps_per_sec', 'steps', 'timed_regions',
                                                        'region_ms_per_step', 'roofline', 'roofline_other_rollout_kernel', 'other_kernels_avg_ms')}
            side['c5_on_1gpu']['rows'] = 32768
            side['c5_on_1gpu']['from'] = 'child process `%s` after the timed region' % how
        except Exception as e:                       # noqa: BLE001
            side['c5_on_1gpu'] = {'error': repr(e)}

    def roof(kernel, nbytes, flop, ms, n):
        """Both roofs of a rollout sweep.  With the split-fp16 engine the sweeps sit closer to the HBM roof (the activation
        stash) than to the matrix roof, so `bound` is "hbm": achieved = algorithmic bytes per launch / average launch
        duration; the matrix view (executed f16 MFMA flop/s against the f16 dense peak, and the algorithmic flop/s against
        the fp32 MFMA peak it replaced) is reported beside it."""
        sec = ms * 1e-3
        gbs = nbytes * B_PER_GPU / sec / 1e9
        tr = traffic.get(kernel.split('<')[0])
        executed = (flop + 2 * HIDDEN_FLOP_PER_STATE) * B_PER_GPU / sec / 1e12      # hidden layer counted 3x
        algorithmic = flop * B_PER_GPU / sec / 1e12
        return {'kernel': kernel, 'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': gbs / HBM_PEAK_GBS, 'frac_hbm': gbs / HBM_PEAK_GBS, 'frac_f16_mfma': executed / F16_MFMA_PEAK_TFLOPS,
                'traffic': tr, 'traffic_from': traffic_from if tr else None, 'avg_ms': ms, 'launches': n,
                'timed_with': 'HIP events on the launch stream around every %d-th launch of the timed region' % PROF_EVERY,
                'algorithmic_bytes_per_launch': nbytes * B_PER_GPU,
                'traffic_gbs': (tr / sec / 1e9) if tr else None,
                'mfma_view': {'executed_f16_tflops': executed, 'peak_f16_tflops': F16_MFMA_PEAK_TFLOPS,
                              'frac_f16': executed / F16_MFMA_PEAK_TFLOPS,
                              'algorithmic_tflops': algorithmic, 'peak_fp32_mfma_tflops': FP32_MFMA_PEAK_TFLOPS,
                              'algorithmic_over_fp32_mfma_peak': algorithmic / FP32_MFMA_PEAK_TFLOPS,
                              'algorithmic_flop_per_launch': flop * B_PER_GPU}}
    r_fwd = roof('k_rollout_fwd<PathTracking>', FWD_BYTES_PER_STATE, FWD_FLOP_PER_STATE, fwd_ms, fwd_n)
    r_bwd = roof('k_rollout_bwd<PathTracking>', BWD_BYTES_PER_STATE, BWD_FLOP_PER_STATE, bwd_ms, bwd_n)
    dominant, other = (r_bwd, r_fwd) if bwd_ms >= fwd_ms else (r_fwd, r_bwd)   # the dominant kernel of the step
    # scalars the driver's record must keep (it retains the flat fields of `roofline`, `config` and `cpu_baseline`; VERDICT r5 missing 4)
    dominant.update({'grad_steps_per_sec': a.steps / dt, 'step_ms': 1e3 * dt / a.steps, 'exact_fp32_ms_per_step': exact.get('ms_per_step'),
                     'other_sweep_kernel': other['kernel'], 'other_sweep_avg_ms': other['avg_ms'], 'other_sweep_frac_hbm': other['frac_hbm']})
    side_flat = {('%s_ms_per_step' % k): v.get('ms_per_step') for k, v in side.items()}
    if side.get('c5_on_1gpu', {}).get('roofline'):
        side_flat['c5_on_1gpu_env_steps_per_sec'] = side['c5_on_1gpu'].get('value')
        side_flat['c5_on_1gpu_dominant_sweep_frac_hbm'] = side['c5_on_1gpu']['roofline'].get('frac_hbm')
        side_flat['c5_on_1gpu_dominant_sweep_avg_ms'] = side['c5_on_1gpu']['roofline'].get('avg_ms')
    out = {
        'metric': 'env-steps/sec + grad-steps/sec, PathTrackingEnv MPG n=25 batch=%d' % B_PER_GPU,
        'value': world * B_PER_GPU * a.steps / dt, 'unit': 'env-steps/s',
        'grad_steps_per_sec': a.steps / dt,
        'model_steps_per_sec': world * B_PER_GPU * N_STEP * a.steps / dt,
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * dt / a.steps,
        # value / ms_per_step / grad_steps_per_sec come from the MEDIAN of `timed_regions` consecutive regions of exactly `steps`
        # steps each (barrier + synchronize on both sides of every region, MAX over ranks per region); all regions listed in order
        'timed_regions': N_REGIONS, 'region_ms_per_step': [1e3 * r / a.steps for r in regions],
        'burn_in_steps': BURN_IN_STEPS,
        'gc': 'gc.collect()+gc.freeze() before the burn-in, gc.disable() inside the timed region',
        'step_ms_median': per_step[len(per_step) // 2], 'step_ms_min': per_step[0], 'step_ms_max': per_step[-1],
        'step_ms_from': 'second pass of the same %d steps, one HIP event per step (not part of value)' % a.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32-via-split-f16' if a.engine == 'split' else 'f32', 'data': 'synthetic',
        'schema': 4,     # 4: value from the median of 5 timed regions, env_step_kernel at a saturating size, side_configs, cpu_baseline legs
                         # 3: roofline.frac = HBM view (round 1: fp32-MFMA view), frac_hbm / frac_f16_mfma under stable keys, exact_fp32_*
        'exact_fp32_ms_per_step': exact.get('ms_per_step'),
        'exact_fp32_from': exact.get('from'),
        'exact_fp32_rollout_kernels_ms': exact.get('rollout_kernels_ms'),
        'engine': a.engine,
        'dtype_note': 'float32 data and accumulation; the 256x256 hidden-layer products run as fp16 hi/lo split operands on the f16 matrix pipe (3 MFMAs per fp32-equivalent step, more accurate than the fp32 fma chain on a single layer; csrc/mlp_core.h)',
        'config': {'workload': 'PathTrackingEnv, MPG-v2 learner, n=25, M=1, %d vectorised envs + replay batch %d per '
                               'GPU; step = worker.sample(%d env-steps) + add_batch + replay + compute_gradient + '
                               '(all-reduce) + apply_gradients' % (B_PER_GPU, B_PER_GPU, B_PER_GPU),
                   'global_batch': world * B_PER_GPU, 'parallelism': 'dp%d' % world,
                   'grad_allreduce_floats': int(learner.flat.numel()), 'native_step_driver': opt._fused is not None,
                   'dist_backend': D.backend(), 'always_exchange': bool(a.always_exchange),
                   'overlap_exchange': os.environ.get('MPG_OVERLAP_EXCHANGE') == '1', 'rows_per_gpu': B_PER_GPU, **side_flat},
        'device': _device_info(),
        'roofline': dominant,
        'roofline_other_rollout_kernel': other,
        'other_kernels_avg_ms': {'k_target_fused': tgt_ms, 'k_critic_fused': crit_ms, 'k_wgrad_multi': wg_ms,
                                 # (one launch when the native driver runs the path-tracking worker: policy pass + env step)
                                 # stable keys (the A/B scripts under tools/ read them): with the fused worker launch the policy pass
                                 # has no launch of its own (0.0) and the env entry is the whole worker launch - see `worker_launch`
                                 'k_forward (worker policy)': pol_ms or 0.0,
                                 'k_step_store_reset (env)': env_ms,
                                 'k_clip_adam_polyak': adam_ms},
        # the ONE exchange step per gradient step (all-reduce of the flat [gradients | statistics] buffer), HIP events on the
        # launch stream around every PROF_EVERY-th exchange of the timed region on rank 0; null on one GPU (none is enqueued)
        'exchange_ms': xch_ms, 'exchange_launches': xch_n,
        # the exchange alone, per form (ms per exchange of the flat buffer; stand-alone, after the timed region; null on one GPU)
        'exchange_forms_ms': exchange_forms,
        'env_step_kernel': {'kernel': 'k_step_store_reset (mpg_env_step_store_reset)', 'avg_ms': step_store_reset_ms, 'launches': 64,
                            'env_steps_per_sec_kernel_only': B_PER_GPU / (step_store_reset_ms * 1e-3),
                            'algorithmic_bytes_per_env_step': 85,
                            'timed_with': 'like env_step_only_kernel (stand-alone; inside the training step the env rides in the worker launch)'},
        # SURVEY section 8d K1 at a saturating size: 85 algorithmic bytes and ~2.3 kflop per env-step; the binding roof is the larger
        # fraction (the kernel is a 20-sub-step dependency chain of sin / cos / atan / divisions per agent: VALU work, not a scan)
        'env_step_kernel_saturating': _env_sat(sat),
        'side_configs': side,
        'worker_launch': {'kernel': 'k_policy_step_store_reset' if not pol_ms else 'k_step_store_reset', 'avg_ms': env_ms, 'launches': env_n},
        # both env rates of SURVEY section 8d under stable keys (kernel-only, 4096 agents per launch)
        'env_steps_per_sec_step_only': B_PER_GPU / (step_only_ms * 1e-3),
        'env_steps_per_sec_step_store_reset': B_PER_GPU / (step_store_reset_ms * 1e-3),
        'env_step_only_kernel': {'kernel': 'k_step (mpg_env_step)', 'avg_ms': step_only_ms, 'launches': 64,
                                 'timed_with': 'one HIP event pair around 64 launches queued behind a spin kernel (back to back on the '
                                               'GPU, launch boundaries included) after the timed region'},
    }
    if world == 1 and not a.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline()
    _flush_c_stdio()
    print(json.dumps(out), flush=True)
    if side_hung:                                # an abandoned collective would block the interpreter's shutdown
        sys.stdout.flush()
        os._exit(0)


if __name__ == '__main__':
    main()
