"""The CPU oracle (oracle/mpg_oracle.py) against golden vectors produced by the unmodified reference
(tests/golden/make_golden.py) and against the reference's own recording mpc/mpc_rl.npy.

Tolerance statement: a float32 implementation `x` is accepted when its error against the
float64 run of the reference graph is at most 4x the error of the reference's own float32 run against
that same float64 run (plus a small floor), and <= 1e-4 relative L2 per gradient array.
"""
import numpy as np
import pytest
import torch

from oracle import mpg_oracle as O


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def net_sizes(cfg):
    od, ad, H = cfg.obs_dim, cfg.act_dim, cfg.H
    pol = od * H + H + H * H + H + H * 2 * ad + 2 * ad
    q = (od + ad) * H + H + H * H + H + H + 1
    return pol, q


def split_grads(flat, cfg, names):
    pol, q = net_sizes(cfg)
    out, o = {}, 0
    for nm in names:
        n = pol if nm == 'policy' else q
        out[nm] = flat[o:o + n]
        o += n
    assert o == flat.size
    return out


# ---- env ------------------------------------------------------------------------------------------
def test_env_step_vs_reference_recording(golden):
    """mpc/mpc_rl.npy: record t+1 = env.step(action_t) applied to record t (obs[0] recorded as v_x, the
    current code uses v_x - 20: SURVEY §4).  198 known-answer transitions, tolerance 5e-6 abs."""
    g = golden('env_step_mpc_rl.npz')
    worst = 0.0
    for who in ('mpc', 'rl'):
        obs, act, rew = g[who + '_obs'].copy(), g[who + '_action'], g[who + '_rew']
        obs[:, 0] -= 20.0
        env = O.PathTrackingEnvOracle(99)
        env.reset(init_obs=obs[:-1].copy())
        o2, r, done, _ = env.step(act[:-1].astype(np.float32))
        err = np.abs(o2 - obs[1:])
        worst = max(worst, err.max())
        assert err.max() <= 5e-6, (who, err.max(0))
        assert np.abs(r - rew[1:]).max() <= 1e-6
        assert done.all()                       # SURVEY B-0: the reference's done flag is always true
    print('max abs err vs mpc_rl.npy', worst)


def test_env_step_vs_reference_run(golden):
    g = golden('env_step_ref.npz')
    env = O.PathTrackingEnvOracle(g['obs0'].shape[0])
    env.reset(init_obs=g['obs0'].copy())
    for t in range(g['actions'].shape[0]):
        o, r, d, _ = env.step(g['actions'][t])
        np.testing.assert_allclose(o, g['obs'][t], rtol=0, atol=2e-6)
        np.testing.assert_allclose(r, g['reward'][t], rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(d.astype(np.uint8), g['done'][t])
        np.testing.assert_allclose(env.veh_full_state, g['full_state'][t], rtol=0, atol=2e-5)


@pytest.mark.parametrize('name,cls', [('model_rollout_ref.npz', O.PathTrackingModelOracle),
                                      ('pendulum_model_ref.npz', O.InvertedPendulumModelOracle)])
def test_model_rollout(golden, name, cls):
    g = golden(name)
    for tag, dt, tol in (('', torch.float32, 2e-5), ('_f64', torch.float64, 1e-12)):
        m = cls()
        m.reset(torch.as_tensor(g['obs0']).to(dt))
        for t in range(g['actions'].shape[0]):
            o, r = m.rollout_out(torch.as_tensor(g['actions'][t]).to(dt), torch.as_tensor(g['eps'][t]).to(dt))
            ref_o, ref_r = g['obs' + tag][t], g['reward' + tag][t]
            scale = 1.0 + np.abs(ref_o)
            assert (np.abs(o.numpy() - ref_o) / scale).max() <= tol, (tag, t)
            assert (np.abs(r.numpy() - ref_r) / (1 + np.abs(ref_r))).max() <= tol, (tag, t)


# ---- learners -------------------------------------------------------------------------------------
def _nets(g, cfg, names, dt):
    return O.Nets(cfg, {k: g['w_' + k] for k in names}, target_scale=g['target_scale'], dtype=dt)


def _check_grads(got_flat, g, key, cfg, names, H):
    ref32, ref64 = g[key], g[key + '_f64']
    got = split_grads(np.concatenate([x.ravel() for x in got_flat]), cfg, names)
    r32 = split_grads(ref32, cfg, names)
    for nm in names:
        e = rel_l2(got[nm], r32[nm])
        assert e <= 1e-4, (key, nm, e)
    if H < 256:          # full float64 tensors are stored only for the small nets
        r64 = split_grads(ref64, cfg, names)
        for nm in names:
            e_ref = rel_l2(r32[nm], r64[nm])
            e_got = rel_l2(got[nm], r64[nm])
            assert e_got <= 4 * e_ref + 1e-6, (key, nm, e_got, e_ref)


@pytest.mark.parametrize('version,H,K', [('v2', 32, 0), ('v1', 32, 0), ('v2', 256, 0), ('v1', 256, 0), ('v2', 256, 3), ('v2', 256, 10)])
def test_mpg_compute_gradient(golden, version, H, K):
    """K = num_future_data (train_script.py:90,146-147): observations carry K look-ahead entries, first layers 6+K / 8+K wide"""
    g = golden('mpg_%s_H%d_B64%s.npz' % (version, H, '_K%d' % K if K else ''))
    cfg = O.Cfg(H=H, obs_dim=6 + K, obs_scale=list(O.OBS_SCALE_PT) + [1.] * K)
    names = ['Q1', 'Q2', 'policy'] if version == 'v2' else ['Q1', 'policy']
    batch = [g['batch_obs'], g['batch_actions'], g['batch_rewards'], g['batch_obs_tp1'], g['batch_dones']]
    for it in (100, 9000):
        nets = _nets(g, cfg, names, torch.float32)
        grads, st = O.mpg_compute_gradient(cfg, nets, batch, g['eps'], it, 'MPG-' + version)
        p = 'it%d_' % it
        np.testing.assert_allclose(st['w_list'], g[p + 'w_list'], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(st['targets'], g[p + 'targets'], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(st['all_losses'], g[p + 'all_losses'], rtol=2e-5, atol=1e-6)
        for k in ('value_mean', 'policy_total_loss', 'policy_gradient_norm', 'q_loss1', 'q_gradient_norm1',
                  'q_loss2', 'q_gradient_norm2'):
            if p + k in g:
                np.testing.assert_allclose(st[k], g[p + k], rtol=5e-5, atol=1e-7, err_msg=k)
        _check_grads(grads, g, p + 'grads', cfg, names, H)
        if it == 100:
            assert rel_l2(st['policy_grad_unclipped'], g['it100_policy_grad_unclipped']) <= 1e-4
    nets = _nets(g, cfg, names, torch.float32)
    td = O.td_error(cfg, nets, *[torch.as_tensor(b) for b in batch[:4]])
    np.testing.assert_allclose(td.numpy(), g['td_error'], rtol=1e-4, atol=2e-6)


def test_mpg_v1_real_env_nstep_rollout(golden):
    g = golden('mpg_v1_H32_B64.npz')
    cfg = O.Cfg(H=32)
    nets = _nets(g, cfg, ['Q1', 'policy'], torch.float32)
    all_r, all_o = O.n_step_env_rollout(cfg, nets, g['batch_obs'], g['batch_actions'])
    np.testing.assert_allclose(all_r, g['nstep_all_rewards'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(all_o[-1], g['nstep_last_obs'], rtol=0, atol=2e-3)


def test_mpg_float64_graph_matches_reference_float64(golden):
    """Same graph in float64: the oracle must agree with the reference's float64 run to ~1e-9, which
    proves the *structure* (which steps carry parameter gradients, stop-gradients, weights) is the same."""
    g = golden('mpg_v2_H32_B64.npz')
    cfg = O.Cfg(H=32)
    names = ['Q1', 'Q2', 'policy']
    batch = [g['batch_obs'], g['batch_actions'], g['batch_rewards'], g['batch_obs_tp1'], g['batch_dones']]
    for it in (100, 9000):
        nets = _nets(g, cfg, names, torch.float64)
        grads, st = O.mpg_compute_gradient(cfg, nets, batch, g['eps'], it)
        got = np.concatenate([x.ravel() for x in grads])
        ref = g['it%d_grads_f64' % it]
        # the fixture stores the float64 run rounded to float32
        assert rel_l2(got, ref) <= 2e-7


@pytest.mark.parametrize('H', [32, 256])
def test_nadp_compute_gradient(golden, H):
    g = golden('nadp_H%d_B64.npz' % H)
    cfg = O.Cfg(env='InvertedPendulumConti-v0', H=H, select=[25], delay_update=1)
    names = ['Q1', 'policy']
    batch = [g['batch_obs'], g['batch_actions']]
    nets = _nets(g, cfg, names, torch.float32)
    grads, st = O.nadp_compute_gradient(cfg, nets, batch, g['eps_q'], g['eps_pi'])
    for k in ('q_loss', 'policy_loss', 'value_mean', 'q_gradient_norm', 'policy_gradient_norm'):
        np.testing.assert_allclose(st[k], g[k], rtol=2e-4, atol=1e-6, err_msg=k)
    _check_grads(grads, g, 'grads', cfg, names, H)


@pytest.mark.parametrize('H', [32, 256])
def test_td3_compute_gradient(golden, H):
    g = golden('td3_H%d_B64.npz' % H)
    cfg = O.Cfg(H=H)
    names = ['Q1', 'Q2', 'policy']
    batch = [g['batch_obs'], g['batch_actions'], g['batch_rewards'], g['batch_obs_tp1'], g['batch_dones']]
    nets = _nets(g, cfg, names, torch.float32)
    grads, st = O.td3_compute_gradient(cfg, nets, batch, g['smooth_eps'])
    np.testing.assert_allclose(st['targets'], g['targets'], rtol=2e-5, atol=2e-6)
    for k in ('q_loss1', 'q_loss2', 'policy_loss', 'value_mean', 'value_var', 'q_gradient_norm1',
              'q_gradient_norm2', 'policy_gradient_norm'):
        np.testing.assert_allclose(st[k], g[k], rtol=1e-4, atol=1e-7, err_msg=k)
    _check_grads(grads, g, 'grads', cfg, names, H)
    nets = _nets(g, cfg, names, torch.float32)
    td = O.td_error(cfg, nets, *[torch.as_tensor(b) for b in batch[:4]])
    np.testing.assert_allclose(td.numpy(), g['td_error'], rtol=1e-4, atol=2e-6)


# ---- segment tree ---------------------------------------------------------------------------------
def test_segment_tree(golden):
    g = golden('segment_tree_ref.npz')
    cap, n, alpha = int(g['capacity']), int(g['n']), float(g['alpha'])
    st = O.SegmentTreeOracle(cap, lambda a, b: a + b, 0.0)
    mt = O.SegmentTreeOracle(cap, min, float('inf'))
    for i, p in enumerate(g['prios']):
        st.set(i, float(p) ** alpha)
        mt.set(i, float(p) ** alpha)
    total = st.reduce(0, n)
    assert total == g['total']                       # bit-exact: same float64 additions in the same tree order
    idx = np.array([st.find_prefixsum_idx(float(x) * total) for x in g['u']])
    np.testing.assert_array_equal(idx, g['idx'])
    for (a, b), s, m in zip(g['ranges'], g['range_sums'], g['range_mins']):
        assert st.reduce(int(a), int(b)) == s and mt.reduce(int(a), int(b)) == m
    assert mt.reduce() == g['min_all']
    for i, p in zip(g['upd_idx'], g['upd_p']):
        st.set(int(i), float(p) ** alpha)
        mt.set(int(i), float(p) ** alpha)
    total2 = st.reduce(0, n)
    assert total2 == g['total2'] and mt.reduce() == g['min_all2']
    idx2 = np.array([st.find_prefixsum_idx(float(x) * total2) for x in g['u']])
    np.testing.assert_array_equal(idx2, g['idx2'])
    np.testing.assert_array_equal([st.find_prefixsum_idx(0.0), st.find_prefixsum_idx(total2)], g['edge_idx'])


# ---- rule-based weights against the reference's own numpy cross-statement --------------------------
def test_rule_based_weights_cross_statement():
    """mpg_learner.py:463-477 restates the rule in numpy (with clip upper bound 1+eta instead of 1.5:
    SURVEY B-8; identical for ite <= total_ite)."""
    sel, eta, T = [0, 25], 0.1, 9000
    for ite in (0, 100, 4499, 4500, 4501, 9000):
        lam = np.clip(1 - eta + 2 * eta * ite / T, 0, 1 + eta)
        b = np.array([lam ** i for i in sel]) if lam < 1 else np.array([(2 - lam) ** (max(sel) - i) for i in sel])
        inv = 1. / (b + 1e-8)
        w = np.exp(inv - inv.max()) / np.exp(inv - inv.max()).sum()
        got = O.rule_based_weights(ite, T, eta, sel).numpy()
        np.testing.assert_allclose(got, w, rtol=2e-4, atol=1e-7)


def test_adam_restatement_cross_checked_against_an_independent_adam():
    """The Keras Adam of policy.py:55-70,127-153 lives in TensorFlow (absent here, version unpinned): the oracle restates
    the published TF ApplyAdam form and says "parity unpinned".  This cross-check against torch.optim.Adam - an
    independent implementation of the same algorithm that only places epsilon differently (inside vs outside the
    bias-corrected denominator) - pins everything else: moments, bias correction, PolynomialDecay schedule, step counter.
    With |g| ~ 1 the epsilon placement changes the update by < 1e-6 relative."""
    rng = np.random.Generator(np.random.PCG64(7))
    n, steps = 500, 40
    sched = (8e-5, 100000, 8e-6)
    w0 = rng.standard_normal(n).astype(np.float32)
    opt = O.AdamState(n)
    w = w0.copy()
    p = torch.nn.Parameter(torch.tensor(w0, dtype=torch.float64))
    for t in range(steps):
        g = rng.standard_normal(n).astype(np.float32)
        lr = O.polynomial_decay(sched, t)
        ref = torch.optim.Adam([p], lr=lr, betas=(0.9, 0.999), eps=1e-7) if t == 0 else ref
        for grp in ref.param_groups:
            grp['lr'] = lr
        p.grad = torch.tensor(g, dtype=torch.float64)
        ref.step()
        w = opt.apply(w, g, sched)
    assert opt.step == steps
    np.testing.assert_allclose(w, p.detach().numpy(), rtol=2e-5, atol=2e-7)
    np.testing.assert_allclose(opt.m, ref.state[p]['exp_avg'].numpy(), rtol=1e-5, atol=1e-7)
    # float32(1) - float32(0.999) = 0.00100004673: the float32 TF functor's second-moment rate differs from the exact
    # 0.001 by 4.7e-5 relative - inherent in the form being restated, not an error of the restatement
    np.testing.assert_allclose(opt.v, ref.state[p]['exp_avg_sq'].numpy(), rtol=1e-4, atol=1e-7)
    # PolynomialDecay(lr0, S, lr_end), power 1: documented closed form
    # (evaluated in float32 like TensorFlow does: within float32 rounding of the real-number value)
    assert abs(O.polynomial_decay(sched, 50000) - (8e-5 - 8e-6) * 0.5 - 8e-6) < 1e-11 and O.polynomial_decay(sched, 10 ** 7) == np.float32(8e-6)
    # ApplyAdam's step size from float32 operands: 6.7e-6 below the real-number formula at t = 1 (float32(0.999) > 0.999)
    import math
    a = O.adam_step_size(sched, 0) / (8e-5 * math.sqrt(1 - 0.999) / (1 - 0.9)) - 1
    assert -7.5e-6 < a < -6e-6, a


# ---- round 2: bench-size cases, the reference's own buffer / evaluator, future-data obs, Q-estimation rollout ---------
from tests import yardstick as Y                          # noqa: E402
from tests.golden_inputs import BENCH_CASES, bench_case_inputs   # noqa: E402


def _flat_nets(nets):
    return {k: np.concatenate([np.asarray(w).ravel() for w in v]).astype(np.float32) for k, v in nets.items()}


@pytest.mark.parametrize('name', sorted(BENCH_CASES))
def test_bench_size_cases(golden, name):
    """C2 (MPG-v2, B = 4096), C3 (NADP, B = 8192), C4 (TD3, B = 65 536): the oracle on the seeded inputs against what the
    reference computed from the same inputs, with the float64 yard-stick on every gradient array."""
    g = golden('bench_%s.npz' % name)
    d = bench_case_inputs(name)
    flat = _flat_nets(d['nets'])
    if d['kind'] == 'MPG-v2':
        cfg, names = O.Cfg(), ['Q1', 'Q2', 'policy']
        nets = O.Nets(cfg, flat, target_scale=g['target_scale'])
        for it in (100, 9000):
            grads, st = O.mpg_compute_gradient(cfg, nets, d['batch'], d['eps'], it, 'MPG-v2')
            p = 'it%d_' % it
            Y.check_gradients(np.concatenate([x.ravel() for x in grads]), g[p + 'grads'], g[p + 'grads_f64'],
                              [('Q1', 8, 1), ('Q2', 8, 1), ('policy', 6, 4)], where=name + ' ' + p)
            Y.check_values(st['targets'][::8], g[p + 'targets_sub'], g[p + 'targets_sub_f64'], what='targets')
            for k in ('value_mean', 'policy_total_loss', 'policy_gradient_norm', 'q_loss1', 'q_gradient_norm1', 'q_loss2',
                      'q_gradient_norm2'):
                np.testing.assert_allclose(st[k], g[p + k], rtol=5e-5, atol=1e-7, err_msg=k)
    elif d['kind'] == 'NADP':
        cfg = O.Cfg(env='InvertedPendulumConti-v0', select=[25], delay_update=1)
        nets = O.Nets(cfg, flat, target_scale=g['target_scale'])
        grads, st = O.nadp_compute_gradient(cfg, nets, d['batch'][:2], d['eps_q'], d['eps_pi'])
        Y.check_gradients(np.concatenate([x.ravel() for x in grads]), g['it0_grads'], g['it0_grads_f64'],
                          [('Q1', 5, 1), ('policy', 4, 2)], where=name)
        for k in ('q_loss', 'policy_loss', 'value_mean', 'q_gradient_norm', 'policy_gradient_norm'):
            np.testing.assert_allclose(st[k], g['it0_' + k], rtol=2e-4, atol=1e-6, err_msg=k)
    else:
        cfg = O.Cfg()
        nets = O.Nets(cfg, flat, target_scale=g['target_scale'])
        grads, st = O.td3_compute_gradient(cfg, nets, d['batch'], d['smooth_eps'])
        Y.check_gradients(np.concatenate([x.ravel() for x in grads]), g['it0_grads'], g['it0_grads_f64'],
                          [('Q1', 8, 1), ('Q2', 8, 1), ('policy', 6, 4)], where=name)
        Y.check_values(st['targets'][::8], g['it0_targets_sub'], g['it0_targets_sub_f64'], what='targets')
        for k in ('q_loss1', 'q_loss2', 'policy_loss', 'value_mean', 'value_var', 'q_gradient_norm1', 'q_gradient_norm2',
                  'policy_gradient_norm'):
            np.testing.assert_allclose(st[k], g['it0_' + k], rtol=1e-4, atol=1e-7, err_msg=k)


def test_trained_networks_case(golden):
    """Round 3: the C2 case on TRAINED networks (tests/golden/trained_weights.npz: 20 000 iterations of the HIP path; outputs by
    the unmodified reference, make_golden.py --only trained_c2) - the oracle restates the reference there as well."""
    g = golden('trained_c2_mpg_v2_B4096.npz')
    tw = golden('trained_weights.npz')
    d = bench_case_inputs('c2_mpg_v2_B4096')
    cfg = O.Cfg()
    nets = O.Nets(cfg, {k: tw['w_' + k] for k in ('Q1', 'Q2', 'policy')}, target_scale=g['target_scale'])
    grads, st = O.mpg_compute_gradient(cfg, nets, d['batch'], d['eps'], 100, 'MPG-v2')
    Y.check_gradients(np.concatenate([x.ravel() for x in grads]), g['it100_grads'], g['it100_grads_f64'],
                      [('Q1', 8, 1), ('Q2', 8, 1), ('policy', 6, 4)], where='trained nets it100', small64=g['it100_grads_small_f64'])
    Y.check_values(st['targets'][::8], g['it100_targets_sub'], g['it100_targets_sub_f64'], what='targets')
    for k in ('value_mean', 'policy_total_loss', 'policy_gradient_norm', 'q_loss1', 'q_gradient_norm1'):
        np.testing.assert_allclose(st[k], g['it100_' + k], rtol=5e-5, atol=1e-7, err_msg=k)


def test_replay_buffer_vs_reference(golden):
    """The reference's own ReplayBuffer (buffer.py imports as-is): ring wrap, storage content, _encode_sample column order
    and dtypes, replay()'s gate and counter - all exact."""
    g = golden('replay_buffer_ref.npz')
    rb = O.ReplayBufferOracle(int(g['capacity']), int(g['replay_starts']))
    o = 0
    for k, m in enumerate(g['sizes']):
        sl = slice(o, o + m)
        rb.add_batch(g['obs'][sl], g['act'][sl], g['rew'][sl], g['obs2'][sl], g['done'][sl].astype(bool))
        o += m
        assert rb._next_idx == g['next_idx'][k] and len(rb) == g['length'][k]
        assert int(rb.replay_gate()) == g['replay_gate'][k]
        enc = rb.encode_sample(g['enc%d_idx' % k])
        for nm, arr in zip(('obs', 'act', 'rew', 'obs2', 'done'), enc):
            ref = g['enc%d_%s' % (k, nm)]
            assert np.array_equal(arr, ref.astype(arr.dtype)) and arr.shape == ref.shape, (k, nm)
        assert list(g['enc%d_dtypes' % k]) == ['float32', 'float32', 'float32', 'float32', 'bool']
    assert rb.replay_times == int(g['replay_times'])
    n = len(rb)
    assert np.array_equal(rb.obs[:n], g['final_obs']) and np.array_equal(rb.act[:n], g['final_act'])
    assert np.array_equal(rb.rew[:n], g['final_rew']) and np.array_equal(rb.obs2[:n], g['final_obs2'])
    assert np.array_equal(rb.done[:n].astype(np.uint8), g['final_done'])


def test_evaluator_metrics_vs_reference(golden):
    """Evaluator.run_n_episodes_parallel + metrics_for_an_episode (evaluator.py:118-184) over 200 closed-loop steps.
    Closed loop, so rounding differences grow along the episode: the allowance is 4x the reference's own
    float32-vs-float64 gap per metric (which is ~1e-5 relative here), not a flat number."""
    g = golden('evaluator_ref.npz')
    cfg = O.Cfg()
    rng = np.random.Generator(np.random.PCG64(0))
    from tests.golden_inputs import mlp_weights_flat
    flat = {'policy': g['w_policy'], 'Q1': mlp_weights_flat(rng, 8, 1), 'Q2': mlp_weights_flat(rng, 8, 1)}
    nets = O.Nets(cfg, flat, target_scale=1.0)
    per, mean = O.run_n_episodes_parallel(cfg, nets, g['init_obs'], int(g['T']))
    keys = [str(k) for k in g['metric_keys']]
    got_mean = np.array([float(mean[k]) for k in keys])
    got_per = np.array([[float(m[k]) for k in keys] for m in per])
    for j, k in enumerate(keys):
        gap = abs(g['mean'][j] - g['mean_f64'][j])
        assert abs(got_mean[j] - g['mean_f64'][j]) <= 4 * gap + 1e-6 * abs(g['mean_f64'][j]) + 1e-9, (k, got_mean[j], g['mean'][j])
        gap_e = np.abs(g['per_episode'][:, j] - g['per_episode_f64'][:, j])
        assert (np.abs(got_per[:, j] - g['per_episode_f64'][:, j]) <= 4 * gap_e.max() + 1e-6 * np.abs(g['per_episode_f64'][:, j]) + 1e-9).all(), k


def test_env_future_data_vs_reference(golden):
    """num_future_data = 3 (path_tracking_env.py:385-402): obs = 6 base entries + 3 look-ahead delta-y terms."""
    g = golden('env_future_ref.npz')
    K, N = int(g['K']), g['obs0'].shape[0]
    env = O.PathTrackingEnvOracle(N, num_future_data=K)
    env.reset(init_obs=np.concatenate([g['obs0'], np.zeros((N, K), np.float32)], 1))
    for t in range(g['actions'].shape[0]):
        o, r, _, _ = env.step(g['actions'][t])
        assert o.shape == (N, 6 + K)
        np.testing.assert_allclose(o, g['obs'][t], rtol=0, atol=5e-6 * (1 + np.abs(g['obs'][t]).max()))
        np.testing.assert_allclose(r, g['reward'][t], rtol=2e-6, atol=1e-5)
    # reset() branch: the future columns as a function of the drawn state
    env2 = O.PathTrackingEnvOracle(N, num_future_data=K)
    fs = g['reset_full_state'].copy()
    vs = fs.copy()
    vs[:, 4] = fs[:, 4] - O.path_phi(fs[:, 5])
    vs[:, 3] = fs[:, 3] - O.path_y(fs[:, 5])
    np.testing.assert_allclose(env2._get_obs(vs, fs), g['reset_obs'], rtol=0, atol=2e-5)


def test_q_estimation_rollout_vs_reference(golden):
    """MPGLearner.model_rollout_for_q_estimation (mpg_learner.py:180-224) for M = 1, 2, 3 and several slice lists."""
    g = golden('q_estimation_ref.npz')
    cfg = O.Cfg()
    for M in (1, 2, 3):
        sel = [int(k) for k in g['M%d_select' % M]]
        for dt, tag in ((torch.float32, ''), (torch.float64, '_f64')):
            nets = _nets(g, cfg, ['Q1', 'Q2', 'policy'], dt)
            y = O.model_rollout_for_q_estimation(cfg, nets, torch.as_tensor(g['batch_obs']), torch.as_tensor(g['batch_actions']),
                                                 torch.as_tensor(g['M%d_eps' % M]), sel, M=M).numpy()
            ref = g['M%d_returns%s' % (M, tag)]
            assert y.shape == ref.shape == (len(sel) * g['batch_obs'].shape[0],)
            if tag:
                # the oracle keeps float32-rounded constants (gamma, obs_scale) in its float64 mode, the reference's
                # float64 graph does not: 0.98f^25 alone differs by 5e-7 from 0.98^25
                assert Y.rel_l2(y, ref) <= 1e-6
            else:
                Y.check_values(y, ref, g['M%d_returns_f64' % M], what='M=%d' % M)


def test_cart_pole_env_restatement_is_self_consistent():
    """InvertedPendulumContiOracle is restated from inverted_pendulum_conti.xml, NOT pinned by the reference (MuJoCo is
    absent - DESIGN.md says "parity unpinned" for this row).  What can be checked without MuJoCo: the mass matrix and
    the equations conserve energy when damping and actuation are switched off (RK4, h = 0.02: drift < 5e-5 relative over
    2 s), the upright equilibrium is unstable with the textbook rate sqrt(m g l (M + m) / ((M + m)(I + m l^2) - m^2 l^2)),
    reward / done follow inverted_pendulum_conti.py:12-18."""
    E = O.InvertedPendulumContiOracle
    c = E.constants()
    assert abs(c['M'] - 1000 * (np.pi * 0.01 * 0.2 + 4 / 3 * np.pi * 1e-3)) < 1e-9          # cart capsule
    env = E(3)
    saved = (E.DAMP_X, E.DAMP_TH)
    try:
        E.DAMP_X = E.DAMP_TH = 0.0
        s0 = np.array([[0., 0.1, 0., 0.], [0.2, -0.15, 0.3, -0.2], [0., 0.05, -0.1, 0.4]])
        env.reset(init_obs=s0)

        def energy(s):
            th = s[:, 1] + c['th0']
            v2 = s[:, 2] ** 2 + 2 * c['l'] * np.cos(th) * s[:, 2] * s[:, 3] + (c['l'] * s[:, 3]) ** 2
            return 0.5 * c['M'] * s[:, 2] ** 2 + 0.5 * c['m'] * v2 + 0.5 * c['I'] * s[:, 3] ** 2 + c['m'] * E.G * c['l'] * np.cos(th)
        e0 = energy(s0)
        for _ in range(50):
            s, r, d, _ = env.step(np.zeros(3))
        assert (np.abs(energy(s) - e0) / np.abs(e0)).max() < 5e-5      # RK4 truncation while the pole swings through
    finally:
        E.DAMP_X, E.DAMP_TH = saved
    env = E(1)
    env.reset(init_obs=np.array([[0., 1e-6 - c['th0'], 0., 0.]]))
    E2 = (c['M'] + c['m']) * (c['I'] + c['m'] * c['l'] ** 2) - (c['m'] * c['l']) ** 2
    lam = np.sqrt(c['m'] * E.G * c['l'] * (c['M'] + c['m']) / E2)
    saved = (E.DAMP_X, E.DAMP_TH)
    try:
        E.DAMP_X = E.DAMP_TH = 0.0
        for _ in range(25):                                   # 1 s
            s, r, d, _ = env.step(np.zeros(1))
    finally:
        E.DAMP_X, E.DAMP_TH = saved
    growth = (s[0, 1] + c['th0']) / 1e-6
    assert abs(growth - np.cosh(lam * 1.0)) / np.cosh(lam * 1.0) < 1e-3
    env = E(2)
    env.reset(init_obs=np.array([[2.05, 0.1, 0., 0.], [0., 0., 0., 0.]]))
    s, r, d, _ = env.step(np.array([3.5, -0.2]))              # action clipped to +-3 by ctrlrange
    assert d[0] and not d[1]
    np.testing.assert_allclose(r, -(0.01 * s[:, 0] ** 2 + s[:, 1] ** 2) - 0.1 * (s[:, 2] ** 2 + s[:, 3] ** 2))


def test_philox_restatements_of_the_device_noise_streams_have_the_stated_laws():
    """The host-side restatements of the device's counter-based draws (oracle.model_noise_philox & co.; the GPU tests pin the
    kernels to them, tests/test_noise_gpu.py).  Here, without a GPU: the in-kernel model noise is a standard normal - mean,
    variance, third and fourth moment of 10^6 draws, no correlation between consecutive steps of a trajectory (lag-1 in t),
    between neighbouring trajectories (cross-row), between the Q-target and the policy rollout of one iteration (counters 2k,
    2k+1) nor between iterations - which is what path_tracking_env.py:119 / inverted_pendulum_model.py:61 (tfd.Normal.sample())
    ask for; the replay indices are uniform on [0, n); the cart-pole reset is U(-0.01, 0.01)^4."""
    n, R = 25, 40000
    z = O.model_noise_philox(n, R, 12345, 2 * 77).astype(np.float64)            # 10^6 draws
    N = z.size
    assert abs(z.mean()) < 4 / np.sqrt(N) and abs(z.var() - 1) < 4 * np.sqrt(2 / N)
    assert abs((z ** 3).mean()) < 4 * np.sqrt(15 / N) and abs((z ** 4).mean() - 3) < 4 * np.sqrt(96 / N)
    corr = lambda a, b: float(np.corrcoef(a.ravel(), b.ravel())[0, 1])
    assert abs(corr(z[:-1], z[1:])) < 4 / np.sqrt(N)                              # lag 1 along a trajectory
    assert abs(corr(z[:, :-1], z[:, 1:])) < 4 / np.sqrt(N)                        # neighbouring trajectories
    z2 = O.model_noise_philox(n, R, 12345, 2 * 77 + 1).astype(np.float64)         # the policy rollout of the same iteration
    z3 = O.model_noise_philox(n, R, 12345, 2 * 78).astype(np.float64)             # the next iteration
    z4 = O.model_noise_philox(n, R, 12346, 2 * 77).astype(np.float64)             # another seed
    assert max(abs(corr(z, z2)), abs(corr(z, z3)), abs(corr(z, z4))) < 4 / np.sqrt(N)
    assert abs(z).max() < 5.9                                                     # 24-bit uniforms: |z| <= sqrt(2 ln 2^25)
    # the per-trajectory sum over the horizon (what a 25-step random walk of the pendulum's p sees): variance n
    assert abs(z.sum(0).var() / n - 1) < 4 * np.sqrt(2 / R)
    zz = O.normal_fill_philox(1000001, 3, 9).astype(np.float64)
    assert zz.shape == (1000001,) and abs(zz.mean()) < 4e-3 and abs(zz.var() - 1) < 6e-3
    assert abs(corr(zz[0:-1:2], zz[1::2])) < 6e-3                                  # the cos / sin pair of one Box-Muller draw
    idx = O.uniform_indices_philox(3072, 1 << 18, 5, 1)
    assert idx.min() == 0 and idx.max() == 3071
    cnt = np.bincount(idx, minlength=3072)
    assert abs(cnt.mean() - (1 << 18) / 3072) < 1e-9 and cnt.std() < 1.15 * np.sqrt((1 << 18) / 3072)
    assert not np.array_equal(idx[:64], O.uniform_indices_philox(3072, 64, 5, 2))
    s = O.cart_pole_reset_philox(100000, 1, 4)
    assert s.dtype == np.float32 and s.shape == (100000, 4) and np.abs(s).max() < 0.01
    assert np.abs(s.mean(0)).max() < 1e-4 and np.abs(s.var(0) - 0.02 ** 2 / 12).max() < 1e-6
    e = O.explore_noise_philox(200000, 2, 0.1, 7, 3)
    assert abs(e.std() - 0.1) < 1e-3 and abs(corr(e[:, 0], e[:, 1])) < 1e-2


def test_oracle_config3_loop_is_deterministic_and_follows_the_optimizer_order():
    """tests/c3_loop.py (the oracle side of the whole-loop test of config 3): two runs from the same weights and seed are
    bit-identical, another seed differs, the ring fills to replay_starts first (optimizer.py:310-313) and a sample is taken every
    `sampling_interval` iterations (optimizer.py:332-337)."""
    from tests.c3_loop import OracleConfig3Loop
    from tests.golden_inputs import mlp_weights_flat
    rng = np.random.Generator(np.random.PCG64(0))
    q, p = mlp_weights_flat(rng, 5, 1), mlp_weights_flat(rng, 4, 2)
    kw = dict(num_agent=16, batch_size=64, replay_batch_size=32, replay_starts=100, sampling_interval=2)
    a, b, c = OracleConfig3Loop(q, p, seed=1, **kw), OracleConfig3Loop(q, p, seed=1, **kw), OracleConfig3Loop(q, p, seed=2, **kw)
    assert a.size == 128 and a.env_ctr == 1 + 8                 # two samples of 4 env steps each, one reset draw per step + the first
    for _ in range(3):
        a.step(); b.step(); c.step()
    assert a.size == 128 + 2 * 64 and a.replay_times == 3 and a.counter == 3
    np.testing.assert_array_equal(a.flat()[0], b.flat()[0])
    np.testing.assert_array_equal(a.ring_obs[:a.size], b.ring_obs[:b.size])
    assert not np.array_equal(a.flat()[0], c.flat()[0])
    assert np.isfinite(a.stats['targets']).all() and a.stats['targets'].shape == (32,)
