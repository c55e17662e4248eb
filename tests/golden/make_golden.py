#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the UNMODIFIED reference files.

Container-only (needs /root/reference, which never travels to the GPU box).  The reference's python
files are imported as they lie under /root/reference on top of the torch-CPU stand-ins in
oracle/refshim/ (TensorFlow / TFP / gym are not installable here).  Everything random is an explicit,
seeded input that is stored next to the outputs, so a fixture is pure data: inputs + what the reference
computed from them.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Fixtures (all float32 unless the name ends in _f64, which is the same graph run in float64 from the
same float32-rounded inputs - the error yard-stick of SURVEY.md §8c):
  env_step_mpc_rl.npz    the reference's own recorded trajectory mpc/mpc_rl.npy, re-packed as arrays
  env_step_ref.npz       PathTrackingEnv.reset(init_obs)/step for N agents x T steps (a1-a5)
  model_rollout_ref.npz  PathTrackingModel.rollout_out x 25 with a fixed action sequence + noise (a6)
  pendulum_model_ref.npz InvertedPendulumModel.rollout_out x 25 (a7)
  mpg_{v1,v2}_H{H}_B{B}.npz   MPGLearner.compute_gradient (a11-a17) at iterations 100 and 9000
  mpg_v2_H256_B64_K3.npz      the same with num_future_data = 3 (obs_dim 9, first layers 9 / 11 wide)
  mpg_v2_H256_B64_K10.npz     the same with num_future_data = 10 (obs_dim 16, first layers 16 / 18 wide)
  nadp_H{H}_B{B}.npz     NADPLearner.compute_gradient on the pendulum model (a18)
  td3_H{H}_B{B}.npz      TD3Learner.compute_gradient with recorded smoothing noise (a19)
  segment_tree_ref.npz   SumSegmentTree / MinSegmentTree primitives (a22)
  bench_c2_mpg_v2_B4096.npz, bench_c3_nadp_B8192.npz, bench_c4_td3_B65536.npz
                         the three learners at the BASELINE.json batch sizes; inputs are seeded draws regenerated on
                         both sides (tests/golden_inputs.py), the fixture holds the reference's outputs only
  replay_buffer_ref.npz  the reference's own ReplayBuffer (buffer.py imports as-is): ring wrap, _encode_sample (a21)
  evaluator_ref.npz      Evaluator.run_n_episodes_parallel + metrics_for_an_episode over 200 steps (f2)
  env_future_ref.npz     PathTrackingEnv with num_future_data = 3 (f3)
  q_estimation_ref.npz   MPGLearner.model_rollout_for_q_estimation, M = 1, 2, 3 (f4)
"""
import argparse
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('MPG_REFERENCE', '/root/reference')
sys.path.insert(0, os.path.join(ROOT, 'oracle', 'refshim'))
sys.path.insert(0, REF)
np.int = int                                            # path_tracking_env.py:371 (numpy < 1.24 era)

import torch                                            # noqa: E402
import tensorflow as tf                                 # noqa: E402  (the stand-in)

sys.path.insert(0, ROOT)
from oracle import mpg_oracle as O                      # noqa: E402  (only the Philox restatements of the device's random streams + the cart-pole stand-in of round 6)

OBS_SCALE_PT = [1., 1., 2., 1., 2.4, 1 / 1200]
OBS_SCALE_PD = [0.001, 1 / 3, 0.1, 0.5]


# ------------------------------------------------------------------------------------------------
# seeded inputs
# ------------------------------------------------------------------------------------------------
def orthogonal(rng, rows, cols, gain):
    a = rng.standard_normal((max(rows, cols), min(rows, cols)))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    if rows < cols:
        q = q.T
    return (gain * q[:rows, :cols]).astype(np.float32)


def mlp_weights(rng, din, H, dout, bias_jitter=0.05):
    """Keras-shaped list [W1,b1,W2,b2,W3,b3], kernels (in,out).  Biases get a small non-zero value so
    that bias handling is exercised (the reference initialises them to 0)."""
    return [orthogonal(rng, din, H, np.sqrt(2.)), (bias_jitter * rng.standard_normal(H)).astype(np.float32),
            orthogonal(rng, H, H, np.sqrt(2.)), (bias_jitter * rng.standard_normal(H)).astype(np.float32),
            orthogonal(rng, H, dout, 1.), (bias_jitter * rng.standard_normal(dout)).astype(np.float32)]


def reset_law_obs(rng, n):
    """obs drawn like PathTrackingEnv.reset() (path_tracking_env.py:426-437) but from a seeded Generator."""
    x = rng.uniform(0, 600, n).astype(np.float32)
    dy = rng.normal(0, 1, n).astype(np.float32)
    dphi = rng.normal(0, np.pi / 9, n).astype(np.float32)
    vx = rng.uniform(15, 25, n).astype(np.float32)
    beta = rng.normal(0, 0.15, n).astype(np.float32)
    vy = (vx * np.tan(beta)).astype(np.float32)
    r = rng.normal(0, 0.3, n).astype(np.float32)
    return np.stack([vx - np.float32(20.), vy, r, dy, dphi, x], 1).astype(np.float32)


def mpg_args(version, B, H, env='PathTracking-v0'):
    pt = env == 'PathTracking-v0'
    d = dict(policy_type='PolicyWithQs', buffer_type='normal', env_id=env, num_agent=8, num_future_data=0,
             alg_name='MPG', learner_version=version, sample_num_in_learner=25, M=1,
             deriv_interval_policy=False, num_rollout_list_for_policy_update=[0, 25],
             num_rollout_list_for_q_estimation=[], eta=0.1, rule_based_bias_total_ite=9000,
             gamma=0.98, gradient_clip_norm=3, num_batch_reuse=1,
             batch_size=512, explore_sigma=0.1, max_buffer_size=500000, replay_starts=3000,
             replay_batch_size=B, replay_alpha=0.6, replay_beta=0.4,
             obs_dim=6 if pt else 4, act_dim=2 if pt else 1,
             value_model_cls='MLP', value_num_hidden_layers=2, value_num_hidden_units=H,
             value_hidden_activation='elu', value_lr_schedule=[8e-5, 100000, 8e-6],
             policy_model_cls='MLP', policy_num_hidden_layers=2, policy_num_hidden_units=H,
             policy_hidden_activation='elu', policy_out_activation='tanh' if pt else 'linear',
             policy_lr_schedule=[3e-5, 100000, 3e-6], alpha=None, alpha_lr_schedule=None,
             policy_only=False, double_Q=(version in ('MPG-v2', 'TD3')), target=True, tau=0.005,
             delay_update=2, deterministic_policy=True, action_range=None if pt else 3.,
             obs_ptype='scale', obs_scale=OBS_SCALE_PT if pt else OBS_SCALE_PD, rew_ptype='scale',
             rew_scale=0.01 if pt else 1., rew_shift=0.,
             policy_smoothing_sigma=0.2, policy_smoothing_clip=0.5)
    return argparse.Namespace(**d)


class NoiseStream(object):
    """Feeds recorded N(0,1) draws to the stand-in's tf.random / tfd.Normal, in call order."""

    def __init__(self, arrays):
        self.arrays = list(arrays)
        self.k = 0

    def __call__(self, shape):
        a = self.arrays[self.k]
        self.k += 1
        assert int(np.prod(shape)) == a.size, (shape, a.shape)
        return a.reshape(shape)


TARGET_SCALE = np.float32(0.97)


def add_targets(nets):
    """Target nets = online nets * 0.97 (float32 product), so that target != online is exercised while the
    fixture only has to carry the online tensors (consumers rebuild the targets with the same product)."""
    for k in list(nets):
        nets[k + '_target'] = [(w * TARGET_SCALE).astype(np.float32) for w in nets[k]]


def sub64(x, H):
    """float64-run tensors are the error yard-stick only: keep every 8th element at H=256, as float32."""
    x = np.asarray(x)
    return (x[::8] if (H >= 256 and x.ndim == 1 and x.size > 4096) else x).astype(np.float32)


def set_policy_weights(pwq, nets):
    """nets: dict name -> keras list.  Targets start as copies unless given explicitly."""
    order = [m.name for m in pwq.models] + [m.name for m in pwq.target_models]
    pwq.set_weights([nets[n] for n in order])


def flat(ws):
    return np.concatenate([np.asarray(w, dtype=np.float64).ravel() for w in ws]).astype(
        np.asarray(ws[0]).dtype)


# ------------------------------------------------------------------------------------------------
# fixtures
# ------------------------------------------------------------------------------------------------
def fx_mpc_rl():
    d = np.load(os.path.join(REF, 'mpc', 'mpc_rl.npy'), allow_pickle=True)
    out = {}
    for who in ('mpc', 'rl'):
        out[who + '_obs'] = np.stack([t[who + '_obs'][0] for t in d]).astype(np.float32)
        out[who + '_action'] = np.stack([np.asarray(t[who + '_action'], dtype=np.float32) for t in d])
        out[who + '_rew'] = np.array([np.float32(t[who + '_rew']) for t in d], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, 'env_step_mpc_rl.npz'), **out)


def fx_env_step(N=64, T=6, seed=0):
    from envs_and_models.path_tracking_env import PathTrackingEnv
    rng = np.random.Generator(np.random.PCG64(seed))
    env = PathTrackingEnv(num_agent=N, num_future_data=0)
    obs0 = reset_law_obs(rng, N)
    # a few hand-picked rows to hit the wraps/clips: x near the 1200 m period, large heading error, slow car
    obs0[0, 5] = 1199.5
    obs0[1, 5] = 0.3
    obs0[2, 4] = 3.1
    obs0[3, 0] = -18.5
    obs0[4, 0] = 14.9
    actions = rng.uniform(-1.3, 1.3, (T, N, 2)).astype(np.float32)       # beyond [-1,1]: env clips
    env.reset(init_obs=obs0.copy())
    obs_l, rew_l, done_l, full_l, others_l = [], [], [], [], []
    for t in range(T):
        o, r, dn, _ = env.step(actions[t])
        obs_l.append(o.copy()), rew_l.append(r.copy()), done_l.append(dn.copy())
        full_l.append(env.veh_full_state.copy())
    np.savez_compressed(os.path.join(HERE, 'env_step_ref.npz'), obs0=obs0, actions=actions,
             obs=np.stack(obs_l).astype(np.float32), reward=np.stack(rew_l).astype(np.float32),
             done=np.stack(done_l).astype(np.uint8), full_state=np.stack(full_l).astype(np.float32))


def fx_model_rollout(N=64, T=25, seed=1):
    from envs_and_models.path_tracking_env import PathTrackingModel
    rng = np.random.Generator(np.random.PCG64(seed))
    obs0 = reset_law_obs(rng, N)
    obs0[0, 4] = 3.12
    obs0[1, 0] = 14.95                     # v_x close to the upper clip (35) after +20
    obs0[2, 0] = -18.9                     # v_x close to the lower clip (1)
    actions = rng.uniform(-1, 1, (T, N, 2)).astype(np.float32)
    eps = rng.standard_normal((T, N)).astype(np.float32)
    out = dict(obs0=obs0, actions=actions, eps=eps)
    for tag, dt in (('', torch.float32), ('_f64', torch.float64)):
        tf.set_ref_dtype(dt)
        tf.set_noise_source(NoiseStream(list(eps)))
        m = PathTrackingModel()
        m.reset(tf.constant(obs0))
        obs_l, rew_l = [], []
        for t in range(T):
            o, r = m.rollout_out(tf.constant(actions[t]))
            obs_l.append(o.numpy()), rew_l.append(r.numpy())
        out['obs' + tag] = np.stack(obs_l)
        out['reward' + tag] = np.stack(rew_l)
    tf.set_ref_dtype(torch.float32)
    tf.set_noise_source(None)
    np.savez_compressed(os.path.join(HERE, 'model_rollout_ref.npz'), **out)


def fx_pendulum_model(N=64, T=25, seed=2):
    from envs_and_models.inverted_pendulum_model import InvertedPendulumModel
    rng = np.random.Generator(np.random.PCG64(seed))
    obs0 = (rng.standard_normal((N, 4)) * np.array([0.5, 0.1, 0.5, 0.5])).astype(np.float32)
    actions = rng.uniform(-3, 3, (T, N, 1)).astype(np.float32)
    eps = rng.standard_normal((T, N)).astype(np.float32)
    out = dict(obs0=obs0, actions=actions, eps=eps)
    for tag, dt in (('', torch.float32), ('_f64', torch.float64)):
        tf.set_ref_dtype(dt)
        tf.set_noise_source(NoiseStream(list(eps)))
        m = InvertedPendulumModel()
        m.reset(tf.constant(obs0))
        obs_l, rew_l = [], []
        for t in range(T):
            o, r = m.rollout_out(tf.constant(actions[t]))
            obs_l.append(o.numpy()), rew_l.append(r.numpy())
        out['obs' + tag] = np.stack(obs_l)
        out['reward' + tag] = np.stack(rew_l)
    tf.set_ref_dtype(torch.float32)
    tf.set_noise_source(None)
    np.savez_compressed(os.path.join(HERE, 'pendulum_model_ref.npz'), **out)


def make_replay_batch_pt(rng, B, K=0):
    """(obs, act, RAW reward, obs', done) produced by the reference env itself (worker.py:108-111).  K look-ahead entries
    (num_future_data): one extra env step first, so that obs as well as obs' carry look-ahead entries the env computed."""
    from envs_and_models.path_tracking_env import PathTrackingEnv
    env = PathTrackingEnv(num_agent=B, num_future_data=K)
    obs = reset_law_obs(rng, B)
    act = np.clip(rng.uniform(-1, 1, (B, 2)) + 0.1 * rng.standard_normal((B, 2)), -1.2, 1.2).astype(np.float32)
    if K:
        act0 = rng.uniform(-0.3, 0.3, (B, 2)).astype(np.float32)
        env.reset(init_obs=np.concatenate([obs, np.zeros((B, K), np.float32)], 1))
        obs = env.step(act0)[0].astype(np.float32)
    else:
        env.reset(init_obs=obs.copy())
    obs2, rew, done, _ = env.step(act)
    return [obs, act, rew.astype(np.float32), obs2.astype(np.float32), done.astype(np.float32)]


def fx_mpg(version, H, B, seed, K=0):
    """K: num_future_data (train_script.py:90,146-147 - obs_dim 6 + K, obs_scale padded with ones); fixture ..._K{K}.npz"""
    from learners.mpg_learner import MPGLearner
    from policy import PolicyWithQs
    rng = np.random.Generator(np.random.PCG64(seed))
    args = mpg_args(version, B, H)
    if K:
        args.num_future_data, args.obs_dim, args.obs_scale = K, 6 + K, OBS_SCALE_PT + [1.] * K
    nets = {'policy': mlp_weights(rng, 6 + K, H, 4), 'Q1': mlp_weights(rng, 8 + K, H, 1)}
    if version == 'MPG-v2':
        nets['Q2'] = mlp_weights(rng, 8 + K, H, 1)
    add_targets(nets)
    batch = make_replay_batch_pt(rng, B, K)
    eps = rng.standard_normal((25, B)).astype(np.float32)
    out = dict(batch_obs=batch[0], batch_actions=batch[1], batch_rewards=batch[2], batch_obs_tp1=batch[3],
               batch_dones=batch[4], eps=eps, iterations=np.array([100, 9000]))
    for k, v in nets.items():
        if not k.endswith('_target'):
            out['w_' + k] = flat(v)
    out['target_scale'] = TARGET_SCALE
    for tag, dt in (('', torch.float32), ('_f64', torch.float64)):
        tf.set_ref_dtype(dt)
        learner = MPGLearner(PolicyWithQs, args)
        set_policy_weights(learner.policy_with_value, nets)
        for it in (100, 9000):
            learner.counter = 0
            tf.set_noise_source(NoiseStream(list(eps)))
            grads = learner.compute_gradient(batch, None, None, it)
            st = learner.get_stats()
            p = 'it%d_' % it
            out[p + 'grads' + tag] = flat(grads) if tag == '' else sub64(flat(grads), H)
            out[p + 'targets' + tag] = np.asarray(learner.batch_data['batch_targets'])
            for key in ('value_mean', 'policy_total_loss', 'policy_gradient_norm', 'q_loss1',
                        'q_gradient_norm1', 'q_loss2', 'q_gradient_norm2'):
                if key in st:
                    out[p + key + tag] = np.asarray(st[key])
            out[p + 'w_list' + tag] = np.asarray(st['w_list'])
            out[p + 'all_losses' + tag] = np.asarray(st['all_losses'])
        if version == 'MPG-v1' and tag == '':
            ro = learner.sample(batch[0].astype(np.float32), batch[1].astype(np.float32))
            out['nstep_all_rewards'] = ro['all_rewards']
            out['nstep_last_obs'] = ro['all_obs_tp1'][-1]
        if tag == '':
            # un-clipped pieces, handy when a clipped comparison fails
            tf.set_noise_source(NoiseStream(list(eps)))
            pg, loss, vm, ws, _, al = learner.policy_forward_and_backward(
                learner.batch_data['batch_obs'], tf.convert_to_tensor(100, dtype=tf.float32), None, None)
            out['it100_policy_grad_unclipped'] = flat([g.numpy() for g in pg])
            out['td_error'] = np.asarray(learner.compute_td_error())
    tf.set_ref_dtype(torch.float32)
    tf.set_noise_source(None)
    np.savez_compressed(os.path.join(HERE, 'mpg_%s_H%d_B%d%s.npz' % (version[-2:], H, B, '_K%d' % K if K else '')), **out)


def fx_nadp(H, B, seed):
    from learners.nadp import NADPLearner
    from policy import PolicyWithQs
    rng = np.random.Generator(np.random.PCG64(seed))
    args = mpg_args('NADP', B, H, env='InvertedPendulumConti-v0')
    args.num_rollout_list_for_policy_update = [25]
    args.num_rollout_list_for_q_estimation = [25]
    args.delay_update = 1
    nets = {'policy': mlp_weights(rng, 4, H, 2), 'Q1': mlp_weights(rng, 5, H, 1)}
    add_targets(nets)
    obs = (rng.standard_normal((B, 4)) * np.array([0.5, 0.1, 0.5, 0.5])).astype(np.float32)
    act = rng.uniform(-3, 3, (B, 1)).astype(np.float32)
    batch = [obs, act, np.zeros(B, np.float32), obs.copy(), np.zeros(B, np.float32)]
    eps_q = rng.standard_normal((25, B)).astype(np.float32)      # Q-target rollout draws first (nadp.py:175)
    eps_pi = rng.standard_normal((25, B)).astype(np.float32)
    out = dict(batch_obs=obs, batch_actions=act, eps_q=eps_q, eps_pi=eps_pi)
    for k, v in nets.items():
        if not k.endswith('_target'):
            out['w_' + k] = flat(v)
    out['target_scale'] = TARGET_SCALE
    for tag, dt in (('', torch.float32), ('_f64', torch.float64)):
        tf.set_ref_dtype(dt)
        learner = NADPLearner(PolicyWithQs, args)
        set_policy_weights(learner.policy_with_value, nets)
        tf.set_noise_source(NoiseStream(list(eps_q) + list(eps_pi)))
        grads = learner.compute_gradient(batch, None, None, 0)
        st = learner.get_stats()
        out['grads' + tag] = flat(grads) if tag == '' else sub64(flat(grads), H)
        for key in ('q_loss', 'policy_loss', 'value_mean', 'q_gradient_norm', 'policy_gradient_norm'):
            out[key + tag] = np.asarray(st[key])
    tf.set_ref_dtype(torch.float32)
    tf.set_noise_source(None)
    np.savez_compressed(os.path.join(HERE, 'nadp_H%d_B%d.npz' % (H, B)), **out)


def fx_td3(H, B, seed):
    from learners.td3 import TD3Learner
    from policy import PolicyWithQs
    rng = np.random.Generator(np.random.PCG64(seed))
    args = mpg_args('TD3', B, H)
    nets = {'policy': mlp_weights(rng, 6, H, 4), 'Q1': mlp_weights(rng, 8, H, 1), 'Q2': mlp_weights(rng, 8, H, 1)}
    add_targets(nets)
    batch = make_replay_batch_pt(rng, B)
    smooth = rng.standard_normal((B, 2)).astype(np.float32)
    out = dict(batch_obs=batch[0], batch_actions=batch[1], batch_rewards=batch[2], batch_obs_tp1=batch[3],
               batch_dones=batch[4], smooth_eps=smooth)
    for k, v in nets.items():
        if not k.endswith('_target'):
            out['w_' + k] = flat(v)
    out['target_scale'] = TARGET_SCALE
    for tag, dt in (('', torch.float32), ('_f64', torch.float64)):
        tf.set_ref_dtype(dt)
        learner = TD3Learner(PolicyWithQs, args)
        set_policy_weights(learner.policy_with_value, nets)
        tf.set_noise_source(NoiseStream([smooth]))
        grads = learner.compute_gradient(batch, None, None, 0)
        st = learner.get_stats()
        out['grads' + tag] = flat(grads) if tag == '' else sub64(flat(grads), H)
        out['targets' + tag] = np.asarray(learner.batch_data['batch_targets'])
        for key in ('q_loss1', 'q_loss2', 'policy_loss', 'value_mean', 'value_var', 'q_gradient_norm1',
                    'q_gradient_norm2', 'policy_gradient_norm'):
            out[key + tag] = np.asarray(st[key])
        if tag == '':
            out['td_error'] = np.asarray(learner.compute_td_error())
    tf.set_ref_dtype(torch.float32)
    tf.set_noise_source(None)
    np.savez_compressed(os.path.join(HERE, 'td3_H%d_B%d.npz' % (H, B)), **out)


def fx_segment_tree(seed=5):
    from utils.segment_tree import MinSegmentTree, SumSegmentTree
    rng = np.random.Generator(np.random.PCG64(seed))
    cap, n = 1024, 700
    prios = (np.abs(rng.standard_normal(n)) + 1e-6)
    st, mt = SumSegmentTree(cap), MinSegmentTree(cap)
    for i, p in enumerate(prios):
        st[i] = float(p) ** 0.6
        mt[i] = float(p) ** 0.6
    total = st.sum(0, n)               # buffer.py:141 (inclusive end; leaf n is 0)
    u = rng.uniform(0, 1, 512)
    idx = np.array([st.find_prefixsum_idx(float(x) * total) for x in u], dtype=np.int64)
    ranges = rng.integers(0, n, (64, 2))
    ranges.sort(axis=1)
    sums = np.array([st.sum(int(a), int(b)) for a, b in ranges])
    mins = np.array([mt.min(int(a), int(b)) for a, b in ranges])
    # update a few priorities (buffer.py:166-189) and re-query
    upd_idx = rng.integers(0, n, 100)
    upd_p = np.abs(rng.standard_normal(100)) + 1e-6
    for i, p in zip(upd_idx, upd_p):
        st[int(i)] = float(p) ** 0.6
        mt[int(i)] = float(p) ** 0.6
    total2 = st.sum(0, n)
    idx2 = np.array([st.find_prefixsum_idx(float(x) * total2) for x in u], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'segment_tree_ref.npz'), capacity=cap, n=n, prios=prios, alpha=0.6, u=u,
             total=total, idx=idx, ranges=ranges, range_sums=sums, range_mins=mins, min_all=mt.min(),
             upd_idx=upd_idx, upd_p=upd_p, total2=total2, idx2=idx2, min_all2=mt.min(),
             edge_idx=np.array([st.find_prefixsum_idx(0.0), st.find_prefixsum_idx(total2)]))


# ------------------------------------------------------------------------------------------------
# round 2: bench-size cases, the reference's own ReplayBuffer / Evaluator, future-data observations, Q-estimation rollout
# ------------------------------------------------------------------------------------------------
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from golden_inputs import BENCH_CASES, bench_case_inputs      # noqa: E402  (seeded inputs shared with the GPU tests)


def trained_nets():
    """The ONLINE networks of a 20 000-iteration MPG-v2 run of the HIP path at the reference's default learning rates
    (tools/train_export.py on the GPU box; evaluation return -4365 -> -5), as Keras-shaped lists."""
    z = np.load(os.path.join(HERE, 'trained_weights.npz'))
    dims = {'policy': (6, 4), 'Q1': (8, 1), 'Q2': (8, 1)}
    nets = {}
    for k, (din, dout) in dims.items():
        f, o, ws = z['w_' + k], 0, []
        for shp in [(din, 256), (256,), (256, 256), (256,), (256, dout), (dout,)]:
            n = int(np.prod(shp))
            ws.append(f[o:o + n].reshape(shp).astype(np.float32))
            o += n
        assert o == f.size
        nets[k] = ws
    return nets


def fx_bench_case(name, trained=False):
    """C2 / C3 / C4 at their full batch sizes.  Inputs are pure functions of the seed (tests/golden_inputs.py), so the
    fixture carries only what the reference computed: the clipped gradient list (float32, complete), every 8th element
    of the float64 run (yard-stick), every 8th target, and the scalar statistics.
    trained: the C2 case on TRAINED networks (trained_nets) instead of freshly initialised ones - the split-fp16 engine's
    accuracy on the weights it actually meets (fixture trained_c2_mpg_v2_B4096.npz)."""
    d = bench_case_inputs(name)
    kind, B, H = d['kind'], d['B'], 256
    nets = trained_nets() if trained else dict(d['nets'])
    add_targets(nets)
    out = dict(target_scale=TARGET_SCALE)
    if kind == 'MPG-v2':
        from learners.mpg_learner import MPGLearner as Learner
        args = mpg_args('MPG-v2', B, H)
        noise = lambda: list(d['eps'])
        its = (100, 9000)
        keys = ('value_mean', 'policy_total_loss', 'policy_gradient_norm', 'q_loss1', 'q_gradient_norm1', 'q_loss2',
                'q_gradient_norm2')
    elif kind == 'NADP':
        from learners.nadp import NADPLearner as Learner
        args = mpg_args('NADP', B, H, env='InvertedPendulumConti-v0')
        args.num_rollout_list_for_policy_update = [25]
        args.num_rollout_list_for_q_estimation = [25]
        args.delay_update = 1
        noise = lambda: list(d['eps_q']) + list(d['eps_pi'])
        its = (0,)
        keys = ('q_loss', 'policy_loss', 'value_mean', 'q_gradient_norm', 'policy_gradient_norm')
    else:
        from learners.td3 import TD3Learner as Learner
        args = mpg_args('TD3', B, H)
        noise = lambda: [d['smooth_eps']]
        its = (0,)
        keys = ('q_loss1', 'q_loss2', 'policy_loss', 'value_mean', 'value_var', 'q_gradient_norm1', 'q_gradient_norm2',
                'policy_gradient_norm')
    from policy import PolicyWithQs
    for tag, dt in (('', torch.float32), ('_f64', torch.float64)):
        tf.set_ref_dtype(dt)
        learner = Learner(PolicyWithQs, args)
        set_policy_weights(learner.policy_with_value, nets)
        for it in its:
            learner.counter = 0
            tf.set_noise_source(NoiseStream(noise()))
            grads = learner.compute_gradient(d['batch'], None, None, it)
            st = learner.get_stats()
            p = 'it%d_' % it
            out[p + 'grads' + tag] = flat(grads) if tag == '' else sub64(flat(grads), H)
            if trained and tag == '_f64':      # output-layer biases are too short for the every-8th subsample: keep them whole
                out[p + 'grads_small_f64'] = np.concatenate([np.asarray(x, np.float64).ravel() for x in grads if np.asarray(x).size < 8])
            if 'batch_targets' in learner.batch_data:
                out[p + 'targets_sub' + tag] = np.asarray(learner.batch_data['batch_targets'])[::8].astype(np.float32)
            for key in keys:
                out[p + key + tag] = np.asarray(st[key])
            if 'w_list' in st:
                out[p + 'w_list' + tag] = np.asarray(st['w_list'])
                out[p + 'all_losses' + tag] = np.asarray(st['all_losses'])
    tf.set_ref_dtype(torch.float32)
    tf.set_noise_source(None)
    np.savez_compressed(os.path.join(HERE, ('trained_%s.npz' if trained else 'bench_%s.npz') % name), **out)


def fx_replay_buffer(seed=40):
    """buffer.py:21-91 imported as-is: ring wrap of add_batch, column order and dtypes of _encode_sample, replay()'s
    gate and counter.  Transitions are what OffPolicyWorker.sample hands over: (obs, act, raw reward, obs', done)."""
    import random
    from buffer import ReplayBuffer
    rng = np.random.Generator(np.random.PCG64(seed))
    cap = 50
    args = argparse.Namespace(max_buffer_size=cap, replay_starts=30, replay_batch_size=16, buffer_log_interval=1000)
    rb = ReplayBuffer(args, 0)
    sizes = [20, 7, 23, 30, 45, 5]                     # 130 transitions: wraps the 50-slot ring more than twice
    n = sum(sizes)
    obs = rng.standard_normal((n, 6)).astype(np.float32)
    act = rng.uniform(-1, 1, (n, 2)).astype(np.float32)
    rew = rng.uniform(-30, 0, n).astype(np.float32)
    obs2 = rng.standard_normal((n, 6)).astype(np.float32)
    done = (rng.uniform(0, 1, n) < 0.5)
    out = dict(capacity=cap, sizes=np.array(sizes), obs=obs, act=act, rew=rew, obs2=obs2, done=done.astype(np.uint8),
               replay_starts=30, replay_batch_size=16)
    o = 0
    next_idx, length, gate = [], [], []
    idx_q = rng.integers(0, 20, (len(sizes), 12))
    for k, m in enumerate(sizes):
        rb.add_batch([(obs[i], act[i], rew[i], obs2[i], done[i]) for i in range(o, o + m)])
        o += m
        next_idx.append(rb._next_idx)
        length.append(len(rb))
        random.seed(k)
        r = rb.replay()
        gate.append(0 if r is None else 1)
        idx = (idx_q[k] % len(rb)).astype(np.int32)
        enc = rb.sample_with_idxes(idx)
        for name, arr in zip(('obs', 'act', 'rew', 'obs2', 'done'), enc[:5]):
            out['enc%d_%s' % (k, name)] = np.asarray(arr)
        out['enc%d_idx' % k] = np.asarray(enc[5])
        out['enc%d_dtypes' % k] = np.array([str(np.asarray(a).dtype) for a in enc[:5]])
    out.update(next_idx=np.array(next_idx), length=np.array(length), replay_gate=np.array(gate), replay_times=rb.replay_times,
               final_obs=np.stack([t[0] for t in rb._storage]), final_act=np.stack([t[1] for t in rb._storage]),
               final_rew=np.array([t[2] for t in rb._storage], np.float32),
               final_obs2=np.stack([t[3] for t in rb._storage]), final_done=np.array([t[4] for t in rb._storage], np.uint8))
    np.savez_compressed(os.path.join(HERE, 'replay_buffer_ref.npz'), **out)


def fx_evaluator(N=8, T=200, seed=50, H=256):
    """Evaluator.run_n_episodes_parallel + metrics_for_an_episode (evaluator.py:118-184) on the reference's env and
    policy, called on a bare object that carries exactly the attributes those two methods read (the constructor wants a
    TensorBoard writer and a log directory).  The start states are the env's own reset() draw (np.random seeded)."""
    import types
    from envs_and_models.path_tracking_env import PathTrackingEnv
    from evaluator import Evaluator
    from policy import PolicyWithQs
    from preprocessor import Preprocessor
    rng = np.random.Generator(np.random.PCG64(seed))
    args = mpg_args('MPG-v2', 64, H)
    args.fixed_steps, args.eval_render, args.num_eval_agent = T, False, N
    nets = {'policy': mlp_weights(rng, 6, H, 4), 'Q1': mlp_weights(rng, 8, H, 1), 'Q2': mlp_weights(rng, 8, H, 1)}
    # a tame controller instead of a random one: the mean head is scaled down so that the closed loop stays inside the
    # env's normal operating range for 200 steps (a random policy saturates steering and spins the car)
    nets['policy'][4] = (nets['policy'][4] * np.float32(0.05)).astype(np.float32)
    add_targets(nets)
    out = dict(w_policy=flat(nets['policy']), N=N, T=T)
    for tag, dt in (('', torch.float32), ('_f64', torch.float64)):
        tf.set_ref_dtype(dt)
        ev = types.SimpleNamespace(args=args, env=PathTrackingEnv(num_agent=N, num_future_data=0),
                                   preprocessor=Preprocessor(**vars(args)), policy_with_value=PolicyWithQs(**vars(args)))
        set_policy_weights(ev.policy_with_value, nets)
        ev.metrics_for_an_episode = types.MethodType(Evaluator.metrics_for_an_episode, ev)
        np.random.seed(seed)
        if tag == '':
            probe = PathTrackingEnv(num_agent=N, num_future_data=0)
            out['init_obs'] = probe.reset().astype(np.float32)          # the same draw the evaluated env makes below
            np.random.seed(seed)
        metrics_list, mean = Evaluator.run_n_episodes_parallel(ev, N)
        keys = sorted(mean.keys())
        out['metric_keys'] = np.array(keys)
        out['per_episode' + tag] = np.array([[float(m[k]) for k in keys] for m in metrics_list], np.float64)
        out['mean' + tag] = np.array([float(mean[k]) for k in keys], np.float64)
    tf.set_ref_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, 'evaluator_ref.npz'), **out)


def fx_env_future(N=64, T=6, K=3, seed=60):
    """PathTrackingEnv with num_future_data = K (path_tracking_env.py:385-402): obs = [6 base entries | K look-ahead
    delta-y terms at x + k * v_x * 0.2]; reset(init_obs) / step like fx_env_step."""
    from envs_and_models.path_tracking_env import PathTrackingEnv
    rng = np.random.Generator(np.random.PCG64(seed))
    env = PathTrackingEnv(num_agent=N, num_future_data=K)
    obs0 = reset_law_obs(rng, N)
    obs0[0, 5] = 1199.5
    obs0[1, 5] = 0.3
    obs0[2, 4] = 3.1
    actions = rng.uniform(-1.3, 1.3, (T, N, 2)).astype(np.float32)
    env.reset(init_obs=np.concatenate([obs0, np.zeros((N, K), np.float32)], 1))
    obs_l, rew_l = [], []
    for t in range(T):
        o, r, dn, _ = env.step(actions[t])
        obs_l.append(o.copy()), rew_l.append(r.copy())
    # the reset() branch: obs of freshly drawn agents (np.random seeded) - the future columns are a function of the state
    np.random.seed(seed)
    env2 = PathTrackingEnv(num_agent=N, num_future_data=K)
    reset_obs = env2.reset().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'env_future_ref.npz'), K=K, obs0=obs0, actions=actions,
                        obs=np.stack(obs_l).astype(np.float32), reward=np.stack(rew_l).astype(np.float32),
                        reset_obs=reset_obs, reset_full_state=env2.veh_full_state.astype(np.float32))


def fx_q_estimation(H=256, B=64, seed=70):
    """MPGLearner.model_rollout_for_q_estimation (mpg_learner.py:180-224): model rollout from (s, a_replay), later actions
    from pi_theta, Q1-target bootstrap at every slice, mean over the M copies, the selected slices concatenated."""
    from learners.mpg_learner import MPGLearner
    from policy import PolicyWithQs
    rng = np.random.Generator(np.random.PCG64(seed))
    nets = {'policy': mlp_weights(rng, 6, H, 4), 'Q1': mlp_weights(rng, 8, H, 1), 'Q2': mlp_weights(rng, 8, H, 1)}
    add_targets(nets)
    obs = reset_law_obs(rng, B)
    act = np.clip(rng.uniform(-1, 1, (B, 2)), -1, 1).astype(np.float32)
    out = dict(batch_obs=obs, batch_actions=act, target_scale=TARGET_SCALE)
    for k, v in nets.items():
        if not k.endswith('_target'):
            out['w_' + k] = flat(v)
    for M, sel in ((1, [0, 5, 25]), (2, [25]), (3, [0, 10])):
        eps = rng.standard_normal((max(sel), M * B)).astype(np.float32)
        out['M%d_eps' % M], out['M%d_select' % M] = eps, np.array(sel)
        for tag, dt in (('', torch.float32), ('_f64', torch.float64)):
            tf.set_ref_dtype(dt)
            args = mpg_args('MPG-v2', B, H)
            args.M, args.num_rollout_list_for_q_estimation = M, sel
            learner = MPGLearner(PolicyWithQs, args)
            set_policy_weights(learner.policy_with_value, nets)
            tf.set_noise_source(NoiseStream(list(eps)))
            y = learner.model_rollout_for_q_estimation(tf.constant(obs), tf.constant(act))
            out['M%d_returns%s' % (M, tag)] = np.asarray(y.numpy())
    tf.set_ref_dtype(torch.float32)
    tf.set_noise_source(None)
    np.savez_compressed(os.path.join(HERE, 'q_estimation_ref.npz'), **out)


# ------------------------------------------------------------------------------------------------
# round 6: the reference's OWN loop code, unmodified - PolicyWithQs.apply_gradients (policy.py:123-171), OffPolicyWorker.sample
# (worker.py:91-119), SingleProcessOffPolicyOptimizer.step (optimizer.py:330-362) - on random streams that are the oracle's
# restatements of the DEVICE's Philox draws, so that tests/c2_loop.py / tests/c3_loop.py (CPU) and the device loops (GPU) can be
# compared with what the reference's control flow produced, iteration for iteration.
# ------------------------------------------------------------------------------------------------
sys.path.insert(0, ROOT)
from golden_inputs import LOOP_SEED, NET_DIMS, apply_case_grads, loop_case_weights, split_keras      # noqa: E402


def loop_args(case, **kw):
    """train_script.py:built_MPG_parser / train_script4mujoco.py:built_NADP_parser defaults for the single-process optimizer"""
    if case == 'nadp':
        args = mpg_args('NADP', 256, 256, env='InvertedPendulumConti-v0')
        args.num_rollout_list_for_policy_update, args.num_rollout_list_for_q_estimation = [25], [25]
        args.delay_update, args.num_agent, args.explore_sigma = 1, 1, None
    elif case == 'td3':
        args = mpg_args('TD3', 256, 256)
    elif case == 'v2k3':                     # num_future_data = 3 (train_script.py:90,146-147: obs_dim 6 + K, obs_scale padded with ones)
        args = mpg_args('MPG-v2', 256, 256)
        args.num_future_data, args.obs_dim, args.obs_scale = 3, 9, OBS_SCALE_PT + [1.] * 3
    else:
        args = mpg_args('MPG-' + case, 256, 256)
        if case == 'v1':
            args.num_batch_reuse = 10
    d = tempfile.mkdtemp(prefix='mpg_golden_')
    args.log_dir, args.model_dir = d + '/logs', d + '/models'
    args.worker_log_interval, args.buffer_log_interval = 10 ** 9, 10 ** 9
    args.obs_ptype, args.rew_ptype = 'scale', 'scale'
    for k, v in kw.items():
        setattr(args, k, v)
    return args


def keras_nets(case, flat_by_name):
    return {name: split_keras(flat_by_name[name], din, dout) for name, din, dout in NET_DIMS[case]}


def set_online_and_targets(pwq, case, flat_by_name):
    """online networks AND targets = the given weights (PolicyWithQs.__init__ copies online -> target, policy.py:60,68)"""
    nets = keras_nets(case, flat_by_name)
    for k in list(nets):
        nets[k + '_target'] = nets[k]
    set_policy_weights(pwq, nets)


def flat_models(pwq):
    return (np.concatenate([flat(m.get_weights()) for m in pwq.models]),
            np.concatenate([flat(m.get_weights()) for m in pwq.target_models]))


class DeviceStreams(object):
    """Every random draw of a reference loop run, replaced IN THE GENERATOR PROCESS by the oracle's restatement of the device's
    counter-based Philox stream for the same event (oracle/mpg_oracle.py; keys as mpg_amd derives them from args.seed):
      np.random.uniform / normal   the env's reset law (path_tracking_env.py:426-437: six calls per reset, O.reset_law_draws with
                                   the reset counter) and the worker's exploration noise (worker.py:98, O.explore_noise_philox with
                                   the policy-call counter)
      random.randint               ReplayBuffer.sample_idxes (buffer.py:70-71): O.uniform_indices_philox(len, B, seed, replay_times)
      tf noise source              the model's in-graph noise (path_tracking_env.py:119, inverted_pendulum_model.py:61): row t of
                                   O.model_noise_philox(n, B, seed, counter); TD3's smoothing noise (td3.py:74): O.normal_fill_philox
    `install()` patches the three entry points, `remove()` restores them.  No reference file is edited."""

    def __init__(self, seed, num_agent, sigma, B, n=25, noise_counters=lambda k: [k], smoothing=False):
        self.w_seed, self.rb_seed, self.l_seed = seed * 1000003, seed * 7919, seed + 12345
        self.num_agent, self.sigma, self.B, self.n = num_agent, sigma, B, n
        self.env_ctr = self.noise_ctr = self.randint_calls = self.model_draws = 0
        self.noise_counters, self.smoothing = noise_counters, smoothing
        self.draw = self.idx = self.eps = None
        self.idx_log = []

    # -- numpy --
    def np_uniform(self, low=0.0, high=1.0, size=None):
        n = int(np.prod(size))
        assert n == self.num_agent, (low, high, size)
        if (low, high) == (0, 600):
            self.draw = O.reset_law_draws(n, self.w_seed, self.env_ctr)
            self.env_ctr += 1
            return self.draw['x'].copy()
        assert (low, high) == (15, 25), (low, high)
        return self.draw['vx'].copy()

    def np_normal(self, loc=0.0, scale=1.0, size=None):
        size = tuple(np.atleast_1d(size))
        if len(size) == 2:                                  # worker.py:98
            assert loc == 0 and scale == self.sigma and size[0] == self.num_agent
            z = O.explore_noise_philox(size[0], size[1], self.sigma, self.w_seed, self.noise_ctr)
            self.noise_ctr += 1
            return z
        assert loc == 0 and size == (self.num_agent,)
        key = {1: 'dy', np.pi / 9: 'dphi', 0.15: 'beta', 0.3: 'r'}[scale]
        return self.draw[key].copy()

    # -- stdlib random (buffer.py:71) --
    def randint(self, a, b):
        k = self.randint_calls % self.B
        if k == 0:
            self.idx = O.uniform_indices_philox(b + 1, self.B, self.rb_seed, self.randint_calls // self.B + 1)
            self.idx_log.append(self.idx.copy())
        assert a == 0
        self.randint_calls += 1
        return int(self.idx[k])

    # -- tf.random.normal / tfd.Normal.sample --
    def tf_noise(self, shape):
        if self.smoothing:                                  # one [B, act] draw per compute_gradient (td3.py:74)
            self.model_draws += 1
            return O.normal_fill_philox(int(np.prod(shape)), self.l_seed, self.model_draws).reshape(shape)
        t = self.model_draws % self.n
        call = self.model_draws // self.n                   # rollouts so far: per_call rollouts per compute_gradient
        per_call = len(self.noise_counters(1))
        if t == 0:
            ctr = self.noise_counters(call // per_call + 1)[call % per_call]
            self.eps = O.model_noise_philox(self.n, int(np.prod(shape)), self.l_seed, ctr)
        self.model_draws += 1
        return self.eps[t].reshape(shape)

    def install(self):
        import random
        self._saved = (np.random.uniform, np.random.normal, random.randint)
        np.random.uniform, np.random.normal, random.randint = self.np_uniform, self.np_normal, self.randint
        tf.set_noise_source(self.tf_noise)

    def remove(self):
        import random
        np.random.uniform, np.random.normal, random.randint = self._saved
        tf.set_noise_source(None)


def fx_apply_gradients(n_iter=6):
    """PolicyWithQs.apply_gradients (policy.py:123-171) x n_iter on given gradient lists: which optimizer steps when (delay_update 2:
    the policy and ALL targets move at iterations 0, 2, 4; delay 1: every iteration), the per-optimizer `iterations`, the learning
    rates the PolynomialDecay schedules hand out (policy.py:54-70), the Polyak mix.  Cases: v2 (double_Q, Q1 / Q2 / policy), v1
    (Q1 / policy, delay 2), nadp (pendulum widths, delay 1).  Per iteration: every 64th parameter and target entry; at the end: all (v2) or
    every 4th (v1, nadp); from the float64 run the parameter and target UPDATES, every 4th entry."""
    from policy import PolicyWithQs
    out = dict(n_iter=n_iter)
    for case in ('v2', 'v1', 'nadp'):
        w0, gl = loop_case_weights(case), apply_case_grads(case, n_iter)
        for tag, dt in (('', torch.float32), ('_f64', torch.float64)):
            tf.set_ref_dtype(dt)
            pwq = PolicyWithQs(**vars(loop_args(case)))
            set_online_and_targets(pwq, case, w0)
            snaps, snaps_t, iters = [], [], []
            w0_flat = np.concatenate([w0[name] for name, _, _ in NET_DIMS[case]]).astype(np.float64)
            for it in range(n_iter):
                grads = [a for name, din, dout in NET_DIMS[case] for a in split_keras(gl[it][name], din, dout)]
                pwq.apply_gradients(tf.constant(it, dtype=tf.int32), grads)
                p, t = flat_models(pwq)
                snaps.append(p[::64]), snaps_t.append(t[::64])
                iters.append([o.iterations for o in pwq.optimizers])
            if tag == '':
                every = 1 if case == 'v2' else 4
                out[case + '_params_sub'], out[case + '_targets_sub'] = np.stack(snaps), np.stack(snaps_t)
                out[case + '_opt_iterations'] = np.array(iters)
                out[case + '_params'], out[case + '_targets'], out[case + '_every'] = p[::every], t[::every], every
            else:       # the float64 run is the error yard-stick: its parameter UPDATE (every 4th entry), as float32
                out[case + '_update_f64'] = (p.astype(np.float64) - w0_flat)[::4].astype(np.float32)
                out[case + '_target_update_f64'] = (t.astype(np.float64) - w0_flat)[::4].astype(np.float32)
    tf.set_ref_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, 'apply_gradients_ref.npz'), **out)


def fx_worker_sample(num_agent=8, batch_size=64, calls=2):
    """OffPolicyWorker.sample (worker.py:91-119) x `calls` on the reference's PathTrackingEnv: policy -> + exploration noise -> env.step
    -> 5-tuples -> env.reset() of every agent (done is always true).  Random inputs: DeviceStreams (seed LOOP_SEED)."""
    from policy import PolicyWithQs
    from worker import OffPolicyWorker
    out = dict(num_agent=num_agent, batch_size=batch_size, calls=calls, seed=LOOP_SEED)
    w0 = loop_case_weights('v2')
    for tag, dt in (('', torch.float32), ('_f64', torch.float64)):
        tf.set_ref_dtype(dt)
        args = loop_args('v2', num_agent=num_agent, batch_size=batch_size)
        st = DeviceStreams(LOOP_SEED, num_agent, args.explore_sigma, args.replay_batch_size)
        st.install()
        try:
            worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
            set_online_and_targets(worker.policy_with_value, 'v2', w0)
            data = []
            for _ in range(calls):
                batch, count = worker.sample_with_count()
                assert count == batch_size
                data += batch
        finally:
            st.remove()
        for k, name in enumerate(('obs', 'act', 'rew', 'obs2', 'done')):
            out[name + tag] = np.stack([np.asarray(t[k]) for t in data]).astype(np.float32 if name != 'done' else np.uint8)
        out['final_obs' + tag] = np.asarray(worker.obs, np.float32)
        out['counters' + tag] = np.array([st.env_ctr, st.noise_ctr])
    tf.set_ref_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, 'worker_sample_ref.npz'), **out)


def _register_cart_pole(seed):
    """InvertedPendulumConti-v0 is a MuJoCo environment (inverted_pendulum_conti.py; MuJoCo is absent): the loop fixture steps the
    oracle's closed-form cart-pole instead (oracle.InvertedPendulumContiOracle - parity of THAT env is unpinned)
    behind the reference's own DummyVecEnv (utils/dummy_vec_env.py).  Reset draw k = the device's Philox stream with counter = env
    steps taken so far (the device re-draws after every step and keeps the draw only where done)."""
    import gym

    class CartPole(gym.Env):
        def __init__(self):
            self.sim, self.n_steps = O.InvertedPendulumContiOracle(1, dtype=np.float32), 0

        def reset(self):
            return self.sim.reset(init_obs=O.cart_pole_reset_philox(1, seed * 1000003, self.n_steps))[0]

        def step(self, a):
            obs, rew, done, info = self.sim.step(np.asarray(a, np.float32).reshape(1, 1))
            self.n_steps += 1
            return obs[0], rew[0], bool(done[0]), info
    gym.register('InvertedPendulumConti-v0', CartPole)


def fx_loop(case, n_iter=20):
    """SingleProcessOffPolicyOptimizer (optimizer.py:286-397) at the reference's defaults - OffPolicyWorker (8 agents, 512 transitions
    per sample) / ReplayBuffer (replay_starts 3000, batch 256) / MPGLearner MPG-v2 (case 'v2'), MPG-v1 ('v1': Q1 + policy, the 25-step
    real-env target of the learner's own 256-agent env recomputed every 10th call; 'v2k3': MPG-v2 with num_future_data = 3 - observations
    with three look-ahead entries through env, worker, ring, model and learner), TD3Learner ('td3', uniform replay) or NADPLearner on the
    pendulum model (case 'nadp': 1 agent behind DummyVecEnv) / PolicyWithQs - constructed (fills the ring) and stepped n_iter times (sampling at
    iterations 0 and 10).  All reference classes unmodified; random inputs: DeviceStreams(LOOP_SEED); initial weights:
    golden_inputs.loop_case_weights.  Per iteration: replay indices, learner statistics, per-optimizer counters, per-network update
    norms, every 64th parameter; at the end: all parameters and targets, the ring."""
    from buffer import ReplayBuffer
    from optimizer import SingleProcessOffPolicyOptimizer
    from policy import PolicyWithQs
    from worker import OffPolicyWorker
    dims = 'v2' if case == 'td3' else case                  # (TD3: the MPG-v2 network set and initial weights)
    smoothing = False
    if case == 'nadp':
        from learners.nadp import NADPLearner as Learner
        _register_cart_pole(LOOP_SEED)
        keys = ('q_loss', 'policy_loss', 'value_mean', 'q_gradient_norm', 'policy_gradient_norm')
        counters = lambda k: [2 * k, 2 * k + 1]             # Q-target rollout, then the policy rollout (nadp.py:175,188)
    elif case == 'td3':
        from learners.td3 import TD3Learner as Learner      # learners/td3.py:150-188; the only tf draw is the smoothing noise (:74)
        keys = ('q_loss1', 'q_loss2', 'policy_loss', 'value_mean', 'q_gradient_norm1', 'q_gradient_norm2', 'policy_gradient_norm')
        counters, smoothing = (lambda k: [k]), True
    else:
        from learners.mpg_learner import MPGLearner as Learner
        keys = ('q_loss1', 'q_loss2', 'value_mean', 'policy_total_loss', 'q_gradient_norm1', 'q_gradient_norm2', 'policy_gradient_norm') \
            if case in ('v2', 'v2k3') else ('q_loss1', 'value_mean', 'policy_total_loss', 'q_gradient_norm1', 'policy_gradient_norm')
        counters = lambda k: [k]
    w0 = loop_case_weights(dims)
    w0_flat = np.concatenate([w0[name] for name, _, _ in NET_DIMS[dims]])
    out = dict(n_iter=n_iter, seed=LOOP_SEED)
    for tag, dt in (('', torch.float32), ('_f64', torch.float64)):
        tf.set_ref_dtype(dt)
        args = loop_args(case)
        st = DeviceStreams(LOOP_SEED, args.num_agent, args.explore_sigma, args.replay_batch_size, noise_counters=counters, smoothing=smoothing)
        st.install()
        try:
            worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
            set_online_and_targets(worker.policy_with_value, dims, w0)
            learner = Learner(PolicyWithQs, args)
            rb = ReplayBuffer(args, 0)
            opt = SingleProcessOffPolicyOptimizer(worker, learner, rb, None, args)
            fill = len(rb)
            stats, iters, upd, sub = [], [], [], []
            for it in range(n_iter):
                opt.step()
                s = learner.get_stats()
                stats.append([float(np.asarray(s[k])) for k in keys])
                pwq = worker.policy_with_value
                iters.append([o.iterations for o in pwq.optimizers])
                p, t = flat_models(pwq)
                o, row = 0, []
                for name, _, _ in NET_DIMS[dims]:
                    n = w0[name].size
                    row.append(float(np.linalg.norm(p[o:o + n].astype(np.float64) - w0[name])))
                    o += n
                upd.append(row), sub.append(p[::64])
        finally:
            st.remove()
        out['stats' + tag], out['update_norms' + tag] = np.array(stats), np.array(upd)
        if tag == '':
            out.update(stat_keys=np.array(keys), fill=fill, opt_iterations=np.array(iters), idx=np.stack(st.idx_log).astype(np.int32),
                       params=p, targets=t, params_sub=np.stack(sub), ring_len=len(rb), ring_next=rb._next_idx,
                       num_sampled_steps=opt.num_sampled_steps, replay_times=rb.replay_times, learner_counter=learner.counter,
                       counters=np.array([st.env_ctr, st.noise_ctr, st.randint_calls, st.model_draws]))
            for k, name in enumerate(('obs', 'act', 'rew', 'obs2', 'done')):
                out['ring_' + name] = np.stack([np.asarray(x[k]) for x in rb._storage]).astype(np.float32)
        else:
            out['update_f64'] = (p.astype(np.float64) - w0_flat)[::4].astype(np.float32)       # yard-stick: the float64 run's parameter update
            out['target_update_f64'] = (t.astype(np.float64) - w0_flat)[::4].astype(np.float32)
    tf.set_ref_dtype(torch.float32)
    np.savez_compressed(os.path.join(HERE, 'loop_%s_ref.npz' % case), **out)


def fx_per_buffer(seed=80):
    """PrioritizedReplayBuffer (buffer.py:94-189) - its METHODS, unmodified: add (max priority), _sample_proportional (scripted
    random.random), sample_with_weights_and_idxes (IS weights), update_priorities (sequential: the last duplicate wins; max_priority).
    The shipped CONSTRUCTOR is dead code (it asserts `args.alpha > 0` - an attribute no parser defines - and reads `args.size`,
    SURVEY B-3), so the object is made with __new__ and given exactly what :114-125 evidently intends: alpha / beta from replay_alpha /
    replay_beta, trees of the next power of two above max_buffer_size, max priority 1.  Likewise `add_batch` hands `add` a weight of 0
    (buffer.py:82: priority 0 ** alpha = 0: never drawn); transitions are added through `add(..., None)`, the max-priority path the
    code evidently means.  Sequence: 500 adds -> draw 256 -> update -> 300 adds (wraps the 700-slot ring: old slots back to max
    priority) -> draw -> update (with duplicate indices) -> draw.  Leaves, indices, weights, max priority after every phase."""
    import random
    from buffer import PrioritizedReplayBuffer, ReplayBuffer
    from utils.segment_tree import MinSegmentTree, SumSegmentTree
    rng = np.random.Generator(np.random.PCG64(seed))
    cap, B = 700, 256
    args = argparse.Namespace(max_buffer_size=cap, replay_starts=100, replay_batch_size=B, buffer_log_interval=10 ** 9,
                              replay_alpha=0.6, replay_beta=0.4)
    rb = PrioritizedReplayBuffer.__new__(PrioritizedReplayBuffer)
    ReplayBuffer.__init__(rb, args, 0)
    rb._alpha, rb._beta = args.replay_alpha, args.replay_beta
    it_capacity = 1
    while it_capacity < cap:
        it_capacity *= 2
    rb._it_sum, rb._it_min, rb._max_priority = SumSegmentTree(it_capacity), MinSegmentTree(it_capacity), 1.0
    n_tr = 800
    obs = rng.standard_normal((n_tr, 6)).astype(np.float32)
    act = rng.uniform(-1, 1, (n_tr, 2)).astype(np.float32)
    rew = rng.uniform(-30, 0, n_tr).astype(np.float32)
    u = rng.uniform(0, 1, (3, B))
    td = (rng.standard_normal((2, B)) * 10.0 ** rng.uniform(-4, 1, (2, B))).astype(np.float32)        # signed, four decades
    out = dict(capacity=cap, tree_capacity=it_capacity, B=B, alpha=0.6, beta=0.4, eps=1e-6, obs=obs, act=act, rew=rew, u=u, td=td)

    def leaves():
        return np.array([rb._it_sum[i] for i in range(it_capacity)], np.float64)

    def draw(k):
        stream = iter(u[k])
        saved, random.random = random.random, lambda: float(next(stream))
        try:
            idx = rb.sample_idxes(B)
        finally:
            random.random = saved
        enc = rb.sample_with_weights_and_idxes(idx)
        out['draw%d_idx' % k], out['draw%d_weights' % k] = np.asarray(idx, np.int64), np.asarray(enc[5], np.float64)
        out['draw%d_obs' % k], out['draw%d_rew' % k] = np.asarray(enc[0]), np.asarray(enc[2])
        out['draw%d_total' % k], out['draw%d_min' % k] = rb._it_sum.sum(), rb._it_min.min()
        return idx
    for i in range(500):
        rb.add(obs[i], act[i], rew[i], obs[i], True, None)
    out['leaves_a'] = leaves()
    idx = draw(0)
    rb.update_priorities(idx, [abs(float(x)) + 1e-6 for x in td[0]])
    out['leaves_b'], out['max_priority_b'] = leaves(), rb._max_priority
    for i in range(500, 800):
        rb.add(obs[i], act[i], rew[i], obs[i], True, None)
    out['leaves_c'], out['next_idx_c'], out['len_c'] = leaves(), rb._next_idx, len(rb)
    idx = draw(1)
    idx = np.concatenate([idx[:200], idx[:56]])                 # duplicates: the LAST occurrence wins (sequential loop, :181-187)
    out['update2_idx'] = np.asarray(idx, np.int64)
    rb.update_priorities(idx, [abs(float(x)) + 1e-6 for x in td[1]])
    out['leaves_d'], out['max_priority_d'] = leaves(), rb._max_priority
    draw(2)
    np.savez_compressed(os.path.join(HERE, 'per_buffer_ref.npz'), **out)


ROUND2 = {'replay_buffer': fx_replay_buffer, 'evaluator': fx_evaluator, 'env_future': fx_env_future,
          'q_estimation': fx_q_estimation}
ROUND2.update({n: (lambda n=n: fx_bench_case(n)) for n in BENCH_CASES})
ROUND2['trained_c2'] = lambda: fx_bench_case('c2_mpg_v2_B4096', trained=True)      # round 3
ROUND2['mpg_future'] = lambda: fx_mpg('MPG-v2', 256, 64, seed=12, K=3)               # round 3: num_future_data = 3
ROUND2['mpg_future10'] = lambda: fx_mpg('MPG-v2', 256, 64, seed=13, K=10)           # round 4: num_future_data = 10 (obs_dim 16, critics 18 wide)
ROUND2.update(apply_gradients=fx_apply_gradients, worker_sample=fx_worker_sample,           # round 6: the reference's own loop code
              loop_v2=lambda: fx_loop('v2'), loop_nadp=lambda: fx_loop('nadp'), loop_td3=lambda: fx_loop('td3'), loop_v1=lambda: fx_loop('v1'), per_buffer=fx_per_buffer,
              loop_v2k3=lambda: fx_loop('v2k3'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='', help='comma-separated round-2 fixture names (%s); default: everything' % ', '.join(ROUND2))
    only = [x for x in ap.parse_args().only.split(',') if x]
    if only:
        torch.manual_seed(0)
        for n in only:
            ROUND2[n]()
        return
    torch.manual_seed(0)
    fx_mpc_rl()
    fx_segment_tree()
    fx_env_step()
    fx_model_rollout()
    fx_pendulum_model()
    for version in ('MPG-v2', 'MPG-v1'):
        fx_mpg(version, 32, 64, seed=10)
        fx_mpg(version, 256, 64, seed=11)
    fx_nadp(32, 64, seed=20)
    fx_nadp(256, 64, seed=21)
    fx_td3(32, 64, seed=30)
    fx_td3(256, 64, seed=31)
    for n in ROUND2:
        ROUND2[n]()
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print('%-28s %8.1f KB' % (f, os.path.getsize(os.path.join(HERE, f)) / 1024))


if __name__ == '__main__':
    main()
