"""Configs 3 and 4: NADP on the pendulum model, TD3, prioritized replay - HIP path
against the goldens produced by the unmodified reference (learners/nadp.py, learners/td3.py, utils/segment_tree.py)."""
import numpy as np
import pytest
import torch

from oracle import mpg_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def dev(x, dt=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dt).to(DEV)


def _check_list(grads, ref, pw, tol):
    got = torch.cat([x.reshape(-1) for x in grads]).cpu().numpy()
    assert got.size == ref.size
    o = 0
    for name in pw.names:
        din, dout = pw.dims[name]
        for shp in O.mlp_shapes(din, 256, dout):
            n = int(np.prod(shp))
            if np.linalg.norm(ref[o:o + n]) > 0:
                e = rel_l2(got[o:o + n], ref[o:o + n])
                assert e <= tol, (name, shp, e)
            o += n


def _load(learner, g):
    pw = learner.policy_with_value
    flat = np.concatenate([g['w_' + n] for n in pw.names])
    pw.set_flat(flat, (flat * np.float32(g['target_scale'])).astype(np.float32))
    return pw


def test_pendulum_model_rollout_q_target_and_nadp_gradient_vs_golden(golden):
    from mpg_amd.config import default_args
    from mpg_amd.learners import NADPLearner
    from mpg_amd.policy import PolicyWithQs
    g = golden('nadp_H256_B64.npz')
    args = default_args('NADP', replay_batch_size=64)
    learner = NADPLearner(PolicyWithQs, args)
    pw = _load(learner, g)
    B = g['batch_obs'].shape[0]
    batch = [dev(g['batch_obs']), dev(g['batch_actions']), torch.zeros(B, device=DEV), dev(g['batch_obs']),
             torch.zeros(B, device=DEV)]
    grads = learner.compute_gradient(batch, None, None, 0, eps_q=dev(g['eps_q']), eps_pi=dev(g['eps_pi']))
    st = learner.get_stats()
    for k in ('q_loss', 'policy_loss', 'value_mean', 'q_gradient_norm', 'policy_gradient_norm'):
        np.testing.assert_allclose(st[k], g[k], rtol=3e-4, atol=1e-6, err_msg=k)
    _check_list(grads, g['grads'], pw, 2e-4)
    # the Q target alone against the oracle in float64
    ocfg = O.Cfg(env='InvertedPendulumConti-v0', select=[25], delay_update=1)
    nets = O.Nets(ocfg, {k: g['w_' + k] for k in ('Q1', 'policy')}, target_scale=g['target_scale'], dtype=torch.float64)
    _, ost = O.nadp_compute_gradient(ocfg, nets, [g['batch_obs'], g['batch_actions']], g['eps_q'], g['eps_pi'])
    np.testing.assert_allclose(learner.batch_data['batch_targets'].cpu().numpy(), ost['targets'], rtol=1e-4, atol=1e-5)


def test_nadp_at_config3_batch_is_finite_and_deterministic():
    from mpg_amd.config import default_args
    from mpg_amd.learners import NADPLearner
    from mpg_amd.policy import PolicyWithQs
    B = 8192
    args = default_args('NADP', replay_batch_size=B)
    learner = NADPLearner(PolicyWithQs, args)
    g = torch.Generator(device='cpu').manual_seed(0)
    obs = (torch.randn(B, 4, generator=g) * torch.tensor([0.5, 0.1, 0.5, 0.5])).to(DEV)
    act = ((torch.rand(B, 1, generator=g) * 6) - 3).to(DEV)
    batch = [obs, act, torch.zeros(B, device=DEV), obs, torch.zeros(B, device=DEV)]
    eq, ep = torch.randn(25, B, generator=g).to(DEV), torch.randn(25, B, generator=g).to(DEV)
    g1 = learner.compute_gradient(batch, None, None, 0, eps_q=eq, eps_pi=ep)
    f1 = learner.flat_grad.clone()
    learner.compute_gradient(batch, None, None, 0, eps_q=eq, eps_pi=ep)
    assert torch.isfinite(f1).all() and torch.equal(f1, learner.flat_grad) and len(g1) == 12


def test_td3_compute_gradient_vs_golden(golden):
    from mpg_amd.config import default_args
    from mpg_amd.learners import TD3Learner
    from mpg_amd.policy import PolicyWithQs
    g = golden('td3_H256_B64.npz')
    args = default_args('TD3', replay_batch_size=64)
    learner = TD3Learner(PolicyWithQs, args)
    pw = _load(learner, g)
    batch = [dev(g[k]) for k in ('batch_obs', 'batch_actions', 'batch_rewards', 'batch_obs_tp1', 'batch_dones')]
    grads = learner.compute_gradient(batch, None, None, 0, smooth_eps=dev(g['smooth_eps']))
    st = learner.get_stats()
    for k in ('q_loss1', 'q_loss2', 'policy_loss', 'value_mean', 'q_gradient_norm1', 'q_gradient_norm2',
              'policy_gradient_norm'):
        np.testing.assert_allclose(st[k], g[k], rtol=1e-4, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(st['value_var'], g['value_var'], rtol=2e-3)
    _check_list(grads, g['grads'], pw, 1e-4)
    np.testing.assert_allclose(learner.compute_td_error().cpu().numpy(), g['td_error'], rtol=1e-4, atol=3e-6)


def test_td3_priorities_td_error_finished_in_the_critic_pass(golden):
    """With a prioritized buffer and num_batch_reuse = 1 the learner does not evaluate Q1(s, a) a second time for the
    priorities' td error (td3.py:83-92, handed to the buffer at optimizer.py:351-353): it keeps y1 and finishes
    y1 - Q1(s, a) from the critic-loss pass of compute_gradient.  Same numbers as the reference's own td_error."""
    from mpg_amd.config import default_args
    from mpg_amd.learners import TD3Learner
    from mpg_amd.policy import PolicyWithQs
    g = golden('td3_H256_B64.npz')
    args = default_args('TD3', replay_batch_size=64, buffer_type='priority')
    learner = TD3Learner(PolicyWithQs, args)
    _load(learner, g)
    batch = [dev(g[k]) for k in ('batch_obs', 'batch_actions', 'batch_rewards', 'batch_obs_tp1', 'batch_dones')]
    idx = torch.arange(64, dtype=torch.int32, device=DEV)
    learner.compute_gradient(batch, 'the buffer', idx, 0, smooth_eps=dev(g['smooth_eps']))
    info = learner.get_info_for_buffer()
    assert info['rb'] == 'the buffer' and info['indexes'] is idx
    np.testing.assert_allclose(info['td_error'].cpu().numpy(), g['td_error'], rtol=1e-4, atol=3e-6)
    np.testing.assert_allclose(info['td_error'].cpu().numpy(), learner.compute_td_error().cpu().numpy(), rtol=0, atol=2e-6)


def test_td3_at_config4_batch_runs():
    from mpg_amd.config import default_args
    from mpg_amd.learners import TD3Learner
    from mpg_amd.policy import PolicyWithQs
    from tests.golden_inputs import reset_law_obs
    B = 65536
    args = default_args('TD3', replay_batch_size=B)
    learner = TD3Learner(PolicyWithQs, args)
    rng = np.random.Generator(np.random.PCG64(0))
    batch = [dev(reset_law_obs(rng, B)), dev(rng.uniform(-1, 1, (B, 2))), dev(rng.standard_normal(B)),
             dev(reset_law_obs(rng, B)), torch.ones(B, device=DEV)]
    grads = learner.compute_gradient(batch, None, None, 0)
    assert len(grads) == 18 and torch.isfinite(learner.flat_grad).all()
    n = learner.norms.cpu().numpy()
    assert (n > 0).all() and np.isfinite(n).all()


def test_segment_tree_primitives_bit_exact_vs_reference(golden):
    """SumSegmentTree / MinSegmentTree content, find_prefixsum_idx and range sums: float64, bit-exact."""
    import mpg_amd._lib as L
    g = golden('segment_tree_ref.npz')
    cap, n, alpha = int(g['capacity']), int(g['n']), float(g['alpha'])
    s = torch.empty(2 * cap, dtype=torch.float64, device=DEV)
    m = torch.empty(2 * cap, dtype=torch.float64, device=DEV)
    stamp = torch.empty(cap, dtype=torch.int32, device=DEV)
    L.call('mpg_per_init', L.ptr(s), L.ptr(m), L.ptr(stamp), L.c_int(cap), L.stream())

    def update(idx, prio):
        d_idx, d_prio = dev(idx, torch.int32), dev(prio)        # keep the tensors alive across the call
        L.call('mpg_per_update', L.ptr(s), L.ptr(m), L.ptr(stamp), L.c_int(cap), L.c_int(len(idx)),
               L.ptr(d_idx), L.ptr(d_prio), L.c_double(alpha), L.c_double(0.0), L.ptr(None), L.stream())
        torch.cuda.synchronize()

    def sample(u):
        idx = torch.empty(len(u), dtype=torch.int32, device=DEV)
        w = torch.empty(len(u), dtype=torch.float32, device=DEV)
        d_u = dev(u, torch.float64)
        L.call('mpg_per_sample', L.ptr(s), L.ptr(m), L.c_int(cap), L.c_int(n), L.c_int(len(u)), L.ptr(d_u),
               L.c_u64(0), L.c_u64(0), L.c_double(0.4), L.ptr(idx), L.ptr(w), L.stream())
        return idx.cpu().numpy(), w.cpu().numpy()
    # the kernel computes pow(float(prio), alpha) from a float32 priority: build the reference tree from the same values
    p32 = g['prios'].astype(np.float32)
    st, mt = O.SegmentTreeOracle(cap, lambda a, b: a + b, 0.0), O.SegmentTreeOracle(cap, min, float('inf'))
    for i, p in enumerate(p32):
        st.set(i, float(p) ** alpha)
        mt.set(i, float(p) ** alpha)
    update(np.arange(n), p32)
    dev_leaves = s[cap:cap + n].cpu().numpy()
    np.testing.assert_allclose(dev_leaves, st.v[cap:cap + n], rtol=4e-16)      # pow() may differ in the last bit
    # make the comparison exact from here on: rebuild the oracle from the device leaves
    for i, v in enumerate(dev_leaves):
        st.set(i, float(v))
        mt.set(i, float(v))
    np.testing.assert_array_equal(s.cpu().numpy()[1:], st.v[1:])               # every internal node bit-identical
    np.testing.assert_array_equal(m.cpu().numpy()[1:], mt.v[1:])
    total = st.reduce(0, n)
    assert total == s[1].item()
    idx, w = sample(g['u'])
    np.testing.assert_array_equal(idx, [st.find_prefixsum_idx(float(x) * total) for x in g['u']])
    np.testing.assert_allclose(w, O.per_is_weights(st, mt, idx, n, 0.4), rtol=2e-6)
    # duplicates in one batch: the last one wins (sequential python loop semantics, buffer.py:181-187)
    ui = np.concatenate([g['upd_idx'], g['upd_idx'][:10]])
    up = np.concatenate([g['upd_p'], g['upd_p'][:10] * 3]).astype(np.float32)
    update(ui, up)
    for i, p in zip(ui, up):
        st.set(int(i), float(s[cap + int(i)].item()))
        mt.set(int(i), float(s[cap + int(i)].item()))
    last = {int(i): float(p) for i, p in zip(ui, up)}
    for i, p in last.items():
        np.testing.assert_allclose(s[cap + i].item(), p ** alpha, rtol=4e-16)
    np.testing.assert_array_equal(s.cpu().numpy()[1:], st.v[1:])
    np.testing.assert_array_equal(m.cpu().numpy()[1:], mt.v[1:])
    idx2, _ = sample(g['u'])
    t2 = st.reduce(0, n)
    np.testing.assert_array_equal(idx2, [st.find_prefixsum_idx(float(x) * t2) for x in g['u']])
    e_idx, _ = sample(np.array([0.0, 1.0]))
    np.testing.assert_array_equal(e_idx, [st.find_prefixsum_idx(0.0), st.find_prefixsum_idx(t2)])


def test_touched_path_update_equals_full_rebuild():
    """Small batches into a large tree (n * 64 <= capacity) update only the ancestors of the touched leaves (k_update_paths); the
    tree content must stay bit-identical to the full rebuild's - checked against the oracle's SegmentTree on the device's leaf values,
    with duplicate indices and neighbouring leaves (shared ancestors) in the batch."""
    import mpg_amd._lib as L
    cap, alpha = 1 << 16, 0.6
    s = torch.empty(2 * cap, dtype=torch.float64, device=DEV)
    m = torch.empty(2 * cap, dtype=torch.float64, device=DEV)
    stamp = torch.empty(cap, dtype=torch.int32, device=DEV)
    L.call('mpg_per_init', L.ptr(s), L.ptr(m), L.ptr(stamp), L.c_int(cap), L.stream())
    rng = np.random.default_rng(5)

    def update(idx, prio):
        d_idx, d_prio = dev(idx, torch.int32), dev(prio)
        L.call('mpg_per_update', L.ptr(s), L.ptr(m), L.ptr(stamp), L.c_int(cap), L.c_int(len(idx)),
               L.ptr(d_idx), L.ptr(d_prio), L.c_double(alpha), L.c_double(1e-6), L.ptr(None), L.stream())
        torch.cuda.synchronize()
    n_fill = 50000
    update(np.arange(n_fill), rng.uniform(0.01, 2.0, n_fill).astype(np.float32))            # full rebuild (n * 64 > capacity)
    st, mt = O.SegmentTreeOracle(cap, lambda a, b: a + b, 0.0), O.SegmentTreeOracle(cap, min, float('inf'))
    leaves = s[cap:].cpu().numpy()
    for i in range(n_fill):
        st.set(i, float(leaves[i]))
        mt.set(i, float(leaves[i]))
    np.testing.assert_array_equal(s.cpu().numpy()[1:], st.v[1:])
    for n in (1, 7, 256, 1000):                                                               # touched paths (n * 64 <= capacity)
        idx = rng.integers(0, n_fill, n)
        idx[: n // 4] = idx[n // 2: n // 2 + n // 4]                                          # duplicates
        if n >= 7:
            idx[-3:] = [100, 101, 102]                                                        # neighbours
        update(idx, rng.uniform(0.001, 5.0, n).astype(np.float32))
        leaves = s[cap:].cpu().numpy()
        for i in set(int(k) for k in idx):
            st.set(i, float(leaves[i]))
            mt.set(i, float(leaves[i]))
        np.testing.assert_array_equal(s.cpu().numpy()[1:], st.v[1:])
        np.testing.assert_array_equal(m.cpu().numpy()[1:], mt.v[1:])


def test_prioritized_replay_at_config4_sizes():
    """capacity 2^19, batch 65536: proportional sampling follows the priorities; round trip add -> sample -> update."""
    from mpg_amd.buffer import PrioritizedReplayBuffer
    from mpg_amd.config import default_args
    args = default_args('TD3', max_buffer_size=500000, replay_starts=1000, replay_batch_size=65536, buffer_type='priority')
    rb = PrioritizedReplayBuffer(args, 0)
    assert rb._cap == 1 << 19
    n = 200000
    g = torch.Generator(device='cpu').manual_seed(1)
    batch = (torch.randn(n, 6, generator=g).to(DEV), torch.randn(n, 2, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV),
             torch.randn(n, 6, generator=g).to(DEV), torch.ones(n, dtype=torch.uint8, device=DEV))
    rb.add_batch(batch)
    assert abs(rb._it_sum[1].item() - n) < 1e-6 and rb._it_min[1].item() == 1.0      # all at max priority 1
    s = rb.replay()
    assert len(s) == 7 and s[-1].max().item() < n and torch.allclose(s[5], torch.ones_like(s[5]))
    # give the first 1000 transitions 100x the td error of the rest
    idx = torch.arange(n, device=DEV, dtype=torch.int32)
    td = torch.full((n,), 0.01, device=DEV)
    td[:1000] = -1.0                                                        # signed, like the learners hand over
    rb.update_priorities(idx, td)
    s = rb.replay()
    frac = (s[-1] < 1000).float().mean().item()
    p_hi, p_lo = (1.0 + 1e-6) ** 0.6, (0.01 + 1e-6) ** 0.6
    expect = 1000 * p_hi / (1000 * p_hi + (n - 1000) * p_lo)
    assert abs(frac - expect) < 0.01, (frac, expect)
    w = s[5]
    assert w.max().item() <= 1.0 + 1e-6 and w.min().item() > 0
    np.testing.assert_array_equal(s[0].cpu().numpy(), batch[0][s[-1].long()].cpu().numpy())


def test_prioritized_sample_in_one_launch_equals_sample_plus_gather():
    """Round 5: PrioritizedReplayBuffer.sample draws, weighs and gathers in ONE launch (mpg_per_sample_gather).  Same Philox stream,
    same descent, same copies: indices, IS weights and all five gathered arrays must equal mpg_per_sample + mpg_replay_gather."""
    from mpg_amd.buffer import PrioritizedReplayBuffer
    from mpg_amd.config import default_args
    rng = np.random.Generator(np.random.PCG64(9))
    N, B = 5000, 4096
    args = default_args('TD3', replay_batch_size=B, buffer_type='priority', replay_starts=N, max_buffer_size=8192)
    rb = PrioritizedReplayBuffer(args, 3)
    rb.add_batch((dev(rng.standard_normal((N, 6))), dev(rng.uniform(-1, 1, (N, 2))), dev(rng.standard_normal(N)),
                  dev(rng.standard_normal((N, 6))), torch.as_tensor(rng.integers(0, 2, N).astype(np.uint8)).to(DEV)))
    rb.update_priorities(torch.arange(N, dtype=torch.int32, device=DEV), dev(np.abs(rng.standard_normal(N)) + 1e-3))
    rb.replay_times = 7
    one = rb.sample(B)
    idx = rb.sample_idxes(B)
    two = list(rb._encode_sample(idx)) + [rb._last_weights, idx]
    assert len(one) == len(two) == 7
    for a, b in zip(one, two):
        assert torch.equal(a, b)
    assert one[4].dtype == torch.float32 and set(one[4].unique().tolist()) <= {0.0, 1.0}


def test_config3_end_to_end_worker_ring_nadp_adam():
    """Config 3 end to end on the device: OffPolicyWorker on the analytic cart-pole (64 pendulums per launch;
    the reference steps ONE MuJoCo pendulum behind DummyVecEnv, train_script4mujoco.py:328, with explore_sigma None) -> replay
    ring -> NADPLearner on the pendulum MODEL -> clip / Adam / Polyak, in SingleProcessOffPolicyOptimizer's order, plus the
    Evaluator's pendulum metrics (evaluator.py:185-211).

    The bar comes from the ORACLE's run of the same loop (tools/config3_oracle_run.py: float64 networks and gradients, the
    oracle's cart-pole, the oracle's Adam; curves under profiles/r04_config3_*): 100-step evaluation returns start near -900
    (the pole falls and swings); the oracle holds the pole upright from iteration 250 on (return -0.2 .. -1.4, theta rms 0.01 ..
    0.03 over four seeds), drifts to -1 .. -60 by iteration 750 - and DIVERGES after ~2000 iterations (value mean 3 -> 7000,
    targets positive although every reward is <= 0: NADP's bootstrapped target is unclipped, nadp.py:87-126, and the model it
    plans with is not the environment it acts in).  The device run of the same pair (tools/config3_device_run.py, four seeds)
    reaches -2.8 .. -51 at iteration 250, -16 .. -46 at 1000, has the same excursion at ~2000 (value mean +18) and recovers.
    So the long-run drift is the algorithm, not the device path.  Checked here: within 1000 iterations the
    pole is held near upright - return above -30 with theta rms < 0.08 at one of the checkpoints 250 / 500 / 750 / 1000 (this
    seed: -9.7 at 250) - far from the -900 of the initial policy; NOT claimed: the reference's plotted base score of -2
    (ploter.py:85; its real environment is MuJoCo, whose parity is unpinned)."""
    from mpg_amd.buffer import ReplayBuffer
    from mpg_amd.config import default_args
    from mpg_amd.evaluator import Evaluator
    from mpg_amd.learners import NADPLearner
    from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer
    from mpg_amd.policy import PolicyWithQs
    from mpg_amd.worker import OffPolicyWorker
    args = default_args('NADP', num_agent=64, batch_size=512, replay_batch_size=512, replay_starts=3000, num_eval_agent=16,
                        fixed_steps=100)
    assert args.env_id == 'InvertedPendulumConti-v0' and args.explore_sigma is None
    worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
    learner = NADPLearner(PolicyWithQs, args)
    rb = ReplayBuffer(args, 0)
    opt = SingleProcessOffPolicyOptimizer(worker, learner, rb, None, args, sampling_interval=10)
    assert opt._fused is not None and opt._fused.c.learner_version == 3      # the native step driver (round 4: NADP too)
    ev = Evaluator(PolicyWithQs, args.env_id, args)
    ev.share_policy(worker.policy_with_value)
    m0 = ev.run_evaluation(0)
    assert set(('x_mean', 'theta_var', 'xdot_mse', 'thetadot_mse_25')) <= set(m0)
    best, best_theta = m0['episode_return'], m0['theta_mse']
    for it in range(250, 1001, 250):
        for _ in range(250):
            opt.step()
        m = ev.run_evaluation(it)
        if m['episode_return'] > best:
            best, best_theta = m['episode_return'], m['theta_mse']
    assert len(rb) >= 3000 and rb.obs.shape[1] == 4 and rb.act.shape[1] == 1
    assert m0['episode_return'] < -300 and best > -30 and best_theta < 0.08, (m0['episode_return'], best, best_theta)
    assert worker.policy_with_value.check_status() == 0 and int(worker.policy_with_value.nonfinite.sum().item()) == 0


def test_nadp_training_trajectory_matches_the_oracle_step_by_step():
    """Closed-loop parity of learner + optimizer over SEVERAL iterations ("bisect env / worker / learner on the
    device"): the device NADP learner and the oracle start from the same weights and are fed the same minibatch and the same
    model noise every iteration; after each of 12 iterations of compute_gradient -> clip -> Adam -> Polyak the device's online and
    target parameters must track the oracle's (float32 oracle: nadp.py:87-241 restated + the oracle's Adam).  A wrong sign, a
    missed step counter, a stale target network or a stash mix-up shows up as drift within a few iterations; what is allowed is
    float32 rounding amplified by Adam's normalisation (measured: 1e-6 relative after 12 steps)."""
    from mpg_amd.config import default_args
    from mpg_amd.learners import NADPLearner
    from mpg_amd.policy import PolicyWithQs
    from tests.golden_inputs import mlp_weights_flat
    rng = np.random.Generator(np.random.PCG64(5))
    B, n = 256, 25
    args = default_args('NADP', replay_batch_size=B)
    learner = NADPLearner(PolicyWithQs, args)
    pw = learner.policy_with_value
    w = {'Q1': mlp_weights_flat(rng, 5, 1), 'policy': mlp_weights_flat(rng, 4, 2)}
    flat = np.concatenate([w[k] for k in pw.names])
    pw.set_flat(flat, flat.copy())
    cfg = O.Cfg(env='InvertedPendulumConti-v0', select=[25], delay_update=1)
    tgt = {k: v.copy() for k, v in w.items()}
    opt = {k: O.AdamState(v.size) for k, v in w.items()}
    worst = 0.0
    for it in range(12):
        obs = (rng.standard_normal((B, 4)) * np.array([0.5, 0.1, 0.5, 0.5])).astype(np.float32)
        act = rng.uniform(-3, 3, (B, 1)).astype(np.float32)
        eps_q = rng.standard_normal((n, B)).astype(np.float32)
        eps_pi = rng.standard_normal((n, B)).astype(np.float32)
        batch = [dev(obs), dev(act), dev(np.zeros(B)), dev(obs), dev(np.zeros(B))]
        learner.counter = 0
        learner.compute_gradient(batch, None, None, it, eps_q=dev(eps_q), eps_pi=dev(eps_pi))
        pw.apply_gradients(it, learner.flat_grad)
        nets = O.Nets(cfg, w, flat_targets=tgt, dtype=torch.float32)
        grads, _ = O.nadp_compute_gradient(cfg, nets, [obs, act], eps_q, eps_pi)
        g = {'Q1': np.concatenate([x.ravel() for x in grads[:6]]).astype(np.float32),
             'policy': np.concatenate([x.ravel() for x in grads[6:]]).astype(np.float32)}
        O.apply_gradients(cfg, w, tgt, opt, g, it, ['Q1', 'policy'])
        got, gott = pw.params.cpu().numpy(), pw.targets.cpu().numpy()
        ref, reft = np.concatenate([w[k] for k in pw.names]), np.concatenate([tgt[k] for k in pw.names])
        e = max(rel_l2(got, ref), rel_l2(gott, reft))
        worst = max(worst, e)
        assert e <= 2e-5, (it, e)
    print('NADP 12-step trajectory: worst relative distance to the oracle %.2e' % worst)


def test_worker_nan_is_reported_from_the_device():
    """worker.py:95-107 `judge_is_nan`: a NaN in an observation (or produced by the policy) sets MPG_STATUS_NAN inside the policy
    kernel; OffPolicyWorker.sample reads the word every `nan_check_interval` calls and raises."""
    from mpg_amd._lib import MpgError
    from mpg_amd.config import default_args
    from mpg_amd.policy import PolicyWithQs
    from mpg_amd.worker import OffPolicyWorker
    args = default_args('MPG-v2', num_agent=64, batch_size=64, nan_check_interval=1)
    worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
    worker.sample()                                            # clean
    worker.obs = worker.obs.clone()
    worker.obs[5, 2] = float('nan')
    with pytest.raises(MpgError, match='judge_is_nan'):
        worker.sample()


@pytest.mark.parametrize('alg,per', [('TD3', True), ('TD3', False), ('NADP', False)])
def test_native_step_driver_equals_method_path_for_td3_and_nadp(alg, per):
    """Round 4: mpg_step_begin / mpg_step_end also drive NADP (learner_version 3) and TD3 with uniform or prioritized replay (4):
    sample -> ring add (+ max-priority leaves) -> replay (proportional sampling + gather) -> targets -> critic losses ->
    priority update (optimizer.py:351-353) -> policy gradient -> clip / Adam / Polyak, enqueued from C++ with no framework
    kernel in between (the smoothing noise is the library's Philox stream in both paths).  After the same number of iterations
    the counters, the replay ring, the segment trees and the drawn indices are identical and parameters / Adam moments /
    targets agree to float32 rounding with the method-by-method path."""
    from mpg_amd.buffer import PrioritizedReplayBuffer, ReplayBuffer
    from mpg_amd.config import default_args
    from mpg_amd.learners import NADPLearner, TD3Learner
    from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer
    from mpg_amd.policy import PolicyWithQs
    from mpg_amd.worker import OffPolicyWorker

    def run(fused):
        args = default_args(alg, num_agent=64, batch_size=128, replay_batch_size=96, replay_starts=512, max_buffer_size=1000,
                            buffer_type='priority' if per else 'normal')
        worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
        learner = (NADPLearner if alg == 'NADP' else TD3Learner)(PolicyWithQs, args)
        rb = (PrioritizedReplayBuffer if per else ReplayBuffer)(args, 0)
        opt = SingleProcessOffPolicyOptimizer(worker, learner, rb, None, args, sampling_interval=3, fused=fused)
        assert (opt._fused is not None) == fused
        for _ in range(9):
            opt.step()
        pw = worker.policy_with_value
        st = learner.get_stats()
        torch.cuda.synchronize()
        out = [pw.params.clone(), pw.targets.clone(), pw.m.clone(), pw.v.clone(), rb.obs.clone(), learner.flat.clone()]
        if per:
            out += [rb._it_sum.clone(), rb._max_priority.clone()]
        key = 'q_loss' if alg == 'NADP' else 'q_loss1'
        return out, (dict(pw.opt_steps), rb._next_idx, len(rb), rb.replay_times, worker._noise_ctr, learner.counter, float(st[key]))
    a, ca = run(True)
    b, cb = run(False)
    assert ca[:6] == cb[:6], (ca, cb)
    assert abs(ca[6] - cb[6]) <= 1e-5 * abs(cb[6]) + 1e-7
    if alg == 'TD3':
        assert torch.equal(a[4], b[4])                # ring observations: the same reset-law draws in the same slots
    else:                                             # (the pendulum's episodes continue: its states follow the rounding-different policy)
        assert (a[4] - b[4]).abs().max().item() <= 1e-4
    for x, y in zip(a[:4], b[:4]):
        assert (x - y).abs().max().item() <= 2e-6 * max(1.0, y.abs().max().item())
    assert ((a[5] - b[5]).norm() / b[5].norm()).item() < 1e-4
    if per:
        assert (a[6] - b[6]).abs().max().item() <= 1e-4 * b[6].abs().max().item()     # priorities = |td| of rounding-different nets
        assert abs(a[7].item() - b[7].item()) <= 1e-4 * abs(b[7].item())


@pytest.mark.parametrize('size', ['b4096', 'c4'])
def test_td3_prioritized_loop_teacher_forced_against_the_oracle_loop(size):
    """Config 4 AS BENCHED - TD3Learner + PrioritizedReplayBuffer through the native driver's learner_version 4 (sample -> ring add at
    max priority -> proportional draw + IS weights + gather -> targets -> critic losses -> |td| + eps priorities into the trees ->
    policy gradient -> clip / Adam / Polyak; buffer.py:127-189, optimizer.py:351-353, td3.py:83-92,150-188) - in closed loop against
    tests/c2_loop.py:OracleConfig4Loop on the same Philox inputs from the same initial weights.  Proportional sampling is discontinuous
    in the priorities, so the oracle loop is TEACHER-FORCED: each iteration it consumes the device's drawn indices and checks them.
    b4096: 20 iterations at replay batch 4096 (512 agents, ring of 16 384 slots that wraps at iteration 16: old transitions are
    overwritten at max priority).  c4: 2 iterations at config 4's batch of 65 536.

    Per iteration:
     (a) EXACT, device against itself: the leaves the device drew from (reconstructed bit for bit from its trees before / after the step)
         put through the reference's tree association (O.heap_tree) and descent with the restated in-kernel uniforms
         (O.per_uniform_philox) give the device's indices - all of them, bit for bit; its IS weights follow from the same tree to 2e-6.
     (i) every device index is a valid find_prefixsum_idx answer for the ORACLE loop's own float64 tree within the band that the L1
         distance between the two leaf sets allows (the loop's leaves come from its own float32 |td|); the number of draws its own tree
         would have answered differently is reported (expected ~B / 256) and bounded by 3 %.
     (ii) IS weights, with the common normalisation factor (leaf_min_device / leaf_min_oracle)^beta taken out (bounded by 5e-2 on its own:
         it hangs on the buffer's ONE smallest |td|): 1e-4 relative for rows whose leaf is not tiny (|td| >= 1e-3: below that a float32
         network difference of 1e-7 is a visible fraction of the priority) and 2e-2 for all; signed TD errors 2e-5 absolute-to-scale;
         max_priority 1e-5 relative.
     (iii) parameter update within 1e-3 relative L2 (measured: see the print), parameters and targets within 1e-5."""
    from mpg_amd.buffer import PrioritizedReplayBuffer
    from mpg_amd.config import default_args
    from mpg_amd.learners import TD3Learner
    from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer
    from mpg_amd.policy import PolicyWithQs
    from mpg_amd.worker import OffPolicyWorker
    from tests.c2_loop import OracleConfig4Loop
    seed = 5
    NA, B, RS, CAP, total = {'b4096': (512, 4096, 8192, 16384, 20), 'c4': (512, 65536, 4096, 8192, 2)}[size]
    nthreads = torch.get_num_threads()
    torch.set_num_threads(8)
    args = default_args('TD3', num_agent=NA, batch_size=NA, replay_batch_size=B, replay_starts=RS, max_buffer_size=CAP, seed=seed,
                        init_seed=seed, buffer_type='priority', nan_check_interval=10 ** 9)
    worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
    learner = TD3Learner(PolicyWithQs, args)
    rb = PrioritizedReplayBuffer(args, 0)
    pw = worker.policy_with_value
    init = pw.params.cpu().numpy().copy()
    off = np.cumsum([0] + list(pw.sizes))
    loop = OracleConfig4Loop({n: init[off[i]:off[i + 1]] for i, n in enumerate(pw.names)}, seed=seed, num_agent=NA, batch_size=NA,
                             replay_batch_size=B, replay_starts=RS, capacity=CAP)
    opt = SingleProcessOffPolicyOptimizer(worker, learner, rb, None, args, sampling_interval=1)
    assert opt._fused is not None and opt._fused.c.learner_version == 4 and opt._fused.c.prioritized == 1
    cap = rb._cap
    assert cap == loop.tree_cap and len(rb) == loop.size == RS
    alpha = rb._alpha
    leaves_of = lambda: rb._it_sum[cap:2 * cap].cpu().numpy().copy()
    np.testing.assert_array_equal(leaves_of(), loop.leaves)                     # the fill: every leaf 1.0 ** alpha
    worst_u = worst_p = 0.0
    flips = 0
    from mpg_amd import _lib as L
    p_sum, p_min = torch.empty(4, dtype=torch.float64, device=DEV), torch.empty(4, dtype=torch.float64, device=DEV)
    p_stamp, p_idx = torch.empty(2, dtype=torch.int32, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
    p_max = torch.zeros(1, dtype=torch.float64, device=DEV)
    L.call('mpg_per_init', L.ptr(p_sum), L.ptr(p_min), L.ptr(p_stamp), L.c_int(2), L.stream())

    def device_leaf(priority):
        """pow(max priority, alpha) in float64 exactly as mpg_per_add computes the leaf of a newly added transition: one add into a
        scratch two-leaf tree whose max priority is set to the value"""
        p_max.fill_(priority)
        L.call('mpg_per_add', L.ptr(p_sum), L.ptr(p_min), L.ptr(p_stamp), L.c_int(2), L.c_int(2), L.c_int(0), L.c_int(1),
               L.c_double(alpha), L.ptr(p_max), L.ptr(p_idx), L.stream())
        return float(p_sum[2].item())
    for it in range(total):
        pre, max_p_pre, next_pre = leaves_of(), float(rb._max_priority.item()), rb._next_idx
        opt.step()
        torch.cuda.synchronize()
        post = leaves_of()
        idx = opt._fused.t['idx'].cpu().numpy().astype(np.int64)
        dev_w = opt._fused.t['b_weights'].cpu().numpy()
        sc = opt._fused.scratch
        dev_td = sc[B * 2 + 2 * B:B * 2 + 3 * B].cpu().numpy()                  # perr = y1 - Q1(s, a) (train_step.cpp td3_gradients)
        # (a) the leaves the device drew from: after the step, except the drawn slots (re-written with the new priorities) - those held their
        #     pre-step value, or, if they were added in this iteration, the max-priority leaf every new slot got
        new = (next_pre + np.arange(NA)) % CAP
        drawn = np.unique(idx)
        at_draw = post.copy()
        at_draw[drawn] = pre[drawn]
        fresh = np.setdiff1d(new, drawn)
        new_leaf = device_leaf(max_p_pre)                # the device's own pow(max priority, alpha), from a scratch one-leaf update
        at_draw[np.intersect1d(new, drawn)] = new_leaf
        assert np.all(post[fresh] == new_leaf) and abs(new_leaf - max_p_pre ** alpha) <= 4e-16 * new_leaf
        tree = O.heap_tree(at_draw, np.add)
        u = O.per_uniform_philox(B, rb.seed, rb.replay_times)
        np.testing.assert_array_equal(O.find_prefixsum_idx_batch(tree, u * tree[1]), idx, err_msg='device draw vs its own tree, iteration %d' % it)
        n = len(rb)
        assert idx.max() < n
        mt = O.heap_tree(np.where(np.arange(cap) < n, at_draw, np.inf), np.minimum)
        w_dev_tree = (at_draw[idx] / tree[1] * n) ** (-rb._beta) / ((mt[1] / tree[1] * n) ** (-rb._beta))
        np.testing.assert_allclose(dev_w, w_dev_tree, rtol=2e-6)
        # the oracle loop, teacher-forced
        loop.step(idx)
        assert loop.size == n and loop.replay_times == rb.replay_times and loop.counter == learner.counter
        own = loop.sum_tree[cap:2 * cap]
        dist = float(np.abs(own - at_draw).sum())
        band = 2 * dist + 1e-12 * tree[1]
        cum = np.cumsum(own)
        mass = loop.u * loop.sum_tree[1]
        lo = np.where(idx > 0, cum[np.maximum(idx - 1, 0)], 0.0)
        assert np.all(mass >= lo - band) and np.all(mass <= cum[idx] + band), (it, 'a device index outside the band of the oracle tree')
        f = int((loop.own_idx != idx).sum())
        flips += f
        assert f <= 0.03 * B, (it, f)
        # (ii)
        scale_td = max(1.0, float(np.abs(loop.td).max()))
        assert np.abs(dev_td - loop.td).max() <= 2e-5 * scale_td, (it, np.abs(dev_td - loop.td).max())
        # IS weights: w_i = (p_i N)^-beta / (p_min N)^-beta = (leaf_i / leaf_min)^-beta - the total and N cancel, and the normalisation hangs
        # on the ONE smallest priority of the buffer (a |td| near zero, whose relative distance between two float32 runs is large): the
        # common factor (leaf_min_device / leaf_min_oracle)^beta is taken out, reported and bounded on its own
        leaf_big = own[idx] >= (1e-3) ** alpha
        cfac = (mt[1] / loop.min_leaf) ** rb._beta
        e_w = np.abs(dev_w / (loop.weights * cfac) - 1.0)
        assert e_w[leaf_big].max() <= 1e-4 and e_w.max() <= 2e-2 and abs(cfac - 1.0) <= 5e-2, (it, e_w[leaf_big].max(), e_w.max(), cfac)
        assert abs(float(rb._max_priority.item()) - float(loop.max_priority)) <= 1e-5 * float(loop.max_priority)
        # (iii)
        got, gott = pw.params.cpu().numpy(), pw.targets.cpu().numpy()
        ref, reft = loop.flat()
        e_p, e_t, e_u = rel_l2(got, ref), rel_l2(gott, reft), rel_l2(got - init, ref - init)
        worst_p, worst_u = max(worst_p, e_p, e_t), max(worst_u, e_u)
        print('iteration %2d: %d of %d draws differ from the oracle tree\'s own; leaves L1 distance %.2e of %.2e; IS weights %.1e (all rows %.1e; '
              'normalisation by the smallest leaf: factor 1 %+.1e); update %.1e' % (it, f, B, dist, tree[1], e_w[leaf_big].max(), e_w.max(), cfac - 1, e_u))
        assert e_p <= 1e-5 and e_t <= 1e-5 and e_u <= 1e-3, (it, e_p, e_t, e_u)
    torch.set_num_threads(nthreads)
    print('TD3 + prioritized replay, teacher-forced (%s), %d iterations: parameters %.1e, update %.1e, %d of %d draws flipped' %
          (size, total, worst_p, worst_u, flips, total * B))


def test_device_per_vs_reference_prioritized_buffer(golden):
    """mpg_amd.PrioritizedReplayBuffer (mpg_per_add / _sample / _update behind it) against the reference's OWN PrioritizedReplayBuffer
    methods run unmodified (per_buffer_ref.npz; its dead constructor bypassed, see the generator): 500 transitions at max priority -> draw
    256 with the fixture's uniforms -> |td| + eps priorities -> 300 more transitions through the wrapping 700-slot ring -> draw ->
    priorities with duplicate indices (the last one wins) -> draw.  Leaves within 1 ulp of python's `**` (4e-16), every drawn index
    exact, IS weights 2e-6 (float32 output), max priority EXACT (float64 on the device since round 6: the float32 it used to be put the
    leaves of newly added transitions 1.8e-9 off the reference's), gathered rows exact."""
    from mpg_amd.buffer import PrioritizedReplayBuffer
    from mpg_amd.config import default_args
    g = golden('per_buffer_ref.npz')
    cap, tcap, B = int(g['capacity']), int(g['tree_capacity']), int(g['B'])
    args = default_args('TD3', max_buffer_size=cap, replay_starts=100, replay_batch_size=B, buffer_type='priority')
    rb = PrioritizedReplayBuffer(args, 0)
    assert rb._cap == tcap and abs(rb._alpha - float(g['alpha'])) < 1e-12 and abs(rb._beta - float(g['beta'])) < 1e-12

    def add(lo, hi):
        rb.add_batch((dev(g['obs'][lo:hi]), dev(g['act'][lo:hi]), dev(g['rew'][lo:hi]), dev(g['obs'][lo:hi]),
                      torch.ones(hi - lo, dtype=torch.uint8, device=DEV)))

    def leaves():
        return rb._it_sum[tcap:2 * tcap].cpu().numpy()

    def draw(k):
        idx = rb.sample_idxes(B, u=dev(g['u'][k], torch.float64))
        w = rb._last_weights.cpu().numpy()
        np.testing.assert_array_equal(idx.cpu().numpy(), g['draw%d_idx' % k], err_msg='draw %d' % k)
        np.testing.assert_allclose(w, g['draw%d_weights' % k], rtol=2e-6)
        enc = rb._encode_sample(idx)
        assert np.array_equal(enc[0].cpu().numpy(), g['draw%d_obs' % k]) and np.array_equal(enc[2].cpu().numpy(), g['draw%d_rew' % k])
        assert abs(rb._it_sum[1].item() - float(g['draw%d_total' % k])) <= 1e-13 * float(g['draw%d_total' % k])
        return idx
    add(0, 500)
    np.testing.assert_allclose(leaves(), g['leaves_a'], rtol=4e-16)
    idx = draw(0)
    rb.update_priorities(idx, dev(g['td'][0]))                      # signed TD errors: |.| + eps on the device
    np.testing.assert_allclose(leaves(), g['leaves_b'], rtol=4e-16)
    assert rb._max_priority.item() == float(g['max_priority_b'])
    add(500, 800)
    assert rb._next_idx == int(g['next_idx_c']) and len(rb) == cap
    np.testing.assert_allclose(leaves(), g['leaves_c'], rtol=4e-16)
    draw(1)
    rb.update_priorities(dev(g['update2_idx'], torch.int32), dev(g['td'][1]))
    np.testing.assert_allclose(leaves(), g['leaves_d'], rtol=4e-16)
    assert rb._max_priority.item() == float(g['max_priority_d'])
    draw(2)
