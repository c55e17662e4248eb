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


@pytest.mark.parametrize('version', ['v2', 'v1'])
@pytest.mark.parametrize('H', [32, 256])
def test_mpg_compute_gradient(golden, version, H):
    g = golden('mpg_%s_H%d_B64.npz' % (version, H))
    cfg = O.Cfg(H=H)
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
    assert abs(O.polynomial_decay(sched, 50000) - (8e-5 - 8e-6) * 0.5 - 8e-6) < 1e-12 and O.polynomial_decay(sched, 10 ** 7) == 8e-6
