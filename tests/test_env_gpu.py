"""HIP PathTrackingEnv (mpg_env_*) against the reference's recording, the reference-run fixture and the
oracle.  Tolerances: the kernel mirrors the reference op-by-op (no FMA contraction); what remains is the
last-ulp behaviour of sin/cos/atan over 20 sub-steps -> 5e-6 abs on the small entries (the survey's
figure for mpc_rl.npy), 4 ulp on x (|x| <= 1200, ulp 1.2e-4), and on delta_y / delta_phi additionally
the path slope (<= 0.37) / curvature times that x tolerance."""
import numpy as np
import pytest
import torch

from oracle import mpg_oracle as O

pytestmark = pytest.mark.gpu


def _close_obs(got, ref, atol=5e-6):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    err = np.abs(got - ref)
    tol = np.full(ref.shape, atol)
    tol[..., 5] = 4 * np.spacing(np.abs(ref[..., 5]).astype(np.float32)).astype(np.float64) + atol
    # delta_y = y - path_y(x) and delta_phi inherit an x that may be off by a few ulp: |path slope| <= 0.37
    tol[..., 3] += 0.4 * tol[..., 5]
    tol[..., 4] += 0.02 * tol[..., 5]
    assert (err <= tol).all(), (err.max(0), np.argwhere(err > tol)[:5])


def test_env_step_vs_reference_recording(golden):
    from mpg_amd.envs import PathTrackingEnv
    g = golden('env_step_mpc_rl.npz')
    for who in ('mpc', 'rl'):
        obs, act, rew = g[who + '_obs'].copy(), g[who + '_action'], g[who + '_rew']
        obs[:, 0] -= 20.0
        env = PathTrackingEnv(num_agent=99)
        env.reset(init_obs=torch.as_tensor(obs[:-1]))
        o2, r, done, _ = env.step(torch.as_tensor(act[:-1].astype(np.float32)))
        _close_obs(o2.cpu().numpy(), obs[1:])
        np.testing.assert_allclose(r.cpu().numpy(), rew[1:], rtol=2e-6, atol=1e-6)
        assert bool(done.all())                     # literal reference behaviour (SURVEY B-0)


def test_env_multi_step_vs_reference_run(golden):
    from mpg_amd.envs import PathTrackingEnv
    g = golden('env_step_ref.npz')
    env = PathTrackingEnv(num_agent=g['obs0'].shape[0])
    env.reset(init_obs=torch.as_tensor(g['obs0']))
    for t in range(g['actions'].shape[0]):
        o, r, d, _ = env.step(torch.as_tensor(g['actions'][t]))
        # errors of earlier steps propagate: tolerance grows with t
        _close_obs(o.cpu().numpy(), g['obs'][t], atol=5e-6 * (t + 1))
        np.testing.assert_allclose(r.cpu().numpy(), g['reward'][t], rtol=1e-5, atol=2e-6)
        np.testing.assert_array_equal(d.cpu().numpy(), g['done'][t])
        full = env.veh_full_state.cpu().numpy()
        np.testing.assert_allclose(full[:, :5], g['full_state'][t][:, :5], rtol=0, atol=2e-5 * (t + 1))


@pytest.mark.parametrize('n', [1, 63, 64, 65, 4096, 100003])
def test_env_step_vs_oracle_ragged_sizes(n):
    from mpg_amd.envs import PathTrackingEnv
    rng = np.random.Generator(np.random.PCG64(n))
    ora = O.PathTrackingEnvOracle(n)
    obs0 = ora.reset(rng=rng).copy()
    act = rng.uniform(-1.2, 1.2, (n, 2)).astype(np.float32)
    ora.reset(init_obs=obs0.copy())
    o_ref, r_ref, d_ref, _ = ora.step(act)
    env = PathTrackingEnv(num_agent=n)
    env.reset(init_obs=torch.as_tensor(obs0))
    o, r, d, _ = env.step(torch.as_tensor(act))
    _close_obs(o.cpu().numpy(), o_ref)
    np.testing.assert_allclose(r.cpu().numpy(), r_ref, rtol=1e-5, atol=2e-6)
    np.testing.assert_array_equal(d.cpu().numpy().astype(bool), d_ref)


def test_env_reset_law_philox_matches_oracle_and_distribution():
    from mpg_amd.envs import PathTrackingEnv
    n = 1 << 16
    env = PathTrackingEnv(num_agent=n, seed=1234)
    obs = env.reset().cpu().numpy()
    full_ref, obs_ref = O.reset_law_philox(n, 1234, 0)
    np.testing.assert_allclose(env.veh_full_state.cpu().numpy(), full_ref, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(obs, obs_ref, rtol=2e-5, atol=2e-5)
    # moments of the reset law (path_tracking_env.py:426-437)
    assert abs(obs[:, 5].mean() - 300) < 3 and obs[:, 5].min() > 0 and obs[:, 5].max() < 600
    assert abs(obs[:, 3].std() - 1.0) < 0.02 and abs(obs[:, 4].std() - np.pi / 9) < 0.01
    assert abs((obs[:, 0] + 20).mean() - 20) < 0.05 and abs(obs[:, 2].std() - 0.3) < 0.01
    # second call: done is all ones (reference quirk) -> every agent re-drawn with a fresh counter
    env.step(torch.zeros(n, 2))
    obs2 = env.reset().cpu().numpy()
    assert np.abs(obs2 - obs).max() > 1.0
    _, obs_ref2 = O.reset_law_philox(n, 1234, 1)
    np.testing.assert_allclose(obs2, obs_ref2, rtol=2e-5, atol=2e-5)


def test_env_25_step_rollout_stays_finite_and_matches_oracle_loosely():
    """Size-independent property at the bench size: 25 consecutive steps (the MPG-v1 n-step sampler) stay
    finite, x stays in (0, 1200], phi in (-pi, pi]."""
    from mpg_amd.envs import PathTrackingEnv
    n = 4096
    env = PathTrackingEnv(num_agent=n, seed=7)
    env.reset()
    g = torch.Generator(device='cpu').manual_seed(0)
    for _ in range(25):
        a = (torch.rand(n, 2, generator=g) * 2 - 1) * 0.3
        o, r, d, _ = env.step(a)
    full = env.veh_full_state.cpu().numpy()
    assert np.isfinite(full).all() and np.isfinite(o.cpu().numpy()).all()
    assert (full[:, 5] > 0).all() and (full[:, 5] <= 1200).all()
    assert (full[:, 4] > -np.pi - 1e-6).all() and (full[:, 4] <= np.pi + 1e-6).all()
    assert (full[:, 0] >= 1).all() and (full[:, 0] <= 35).all()


@pytest.mark.parametrize('n,cap,nxt', [(300, 1000, 850), (70001, 100000, 60000)])
def test_fused_step_store_reset_equals_separate_calls(n, cap, nxt):
    """mpg_env_step_store_reset == mpg_env_step + mpg_replay_add + mpg_env_reset, bit for bit, incl. a wrapping ring (850 + 300 and
    60 000 + 70 001 wrap).  n = 70 001: from 65 536 agents on the launch takes its ONE-LANE-per-agent form (k_step_store_reset_1, round 6:
    the four-lane form is built for 4096-agent latency) - the same bits by the same comparison."""
    import mpg_amd._lib as L
    from mpg_amd.envs import PathTrackingEnv
    rng = np.random.Generator(np.random.PCG64(3))
    act = torch.as_tensor(rng.uniform(-1.2, 1.2, (n, 2)).astype(np.float32)).cuda()
    env_a, env_b = PathTrackingEnv(num_agent=n, seed=9), PathTrackingEnv(num_agent=n, seed=9)
    obs0 = env_a.reset().clone()
    env_b.reset()
    # separate path
    o2, r, d, _ = env_a.step(act)
    ring_a = [torch.zeros(cap, 6).cuda(), torch.zeros(cap, 2).cuda(), torch.zeros(cap).cuda(), torch.zeros(cap, 6).cuda(),
              torch.zeros(cap, dtype=torch.uint8).cuda()]
    L.call('mpg_replay_add', L.c_int(cap), L.c_int(nxt), L.c_int(n), L.c_int(6), L.c_int(2), L.ptr(obs0), L.ptr(act), L.ptr(r),
           L.ptr(o2), L.ptr(d), *[L.ptr(t) for t in ring_a], L.stream())
    obs_a = env_a.reset()
    # fused path
    ring_b = [torch.zeros_like(t) for t in ring_a]
    obs_b = torch.empty(n, 6).cuda()
    done_b = torch.empty(n, dtype=torch.uint8).cuda()
    L.call('mpg_env_step_store_reset', L.c_int(0), L.c_int(n), L.c_int(6), L.ptr(env_b._state), L.ptr(act), L.c_int(cap), L.c_int(nxt),
           *[L.ptr(t) for t in ring_b], L.c_u64(env_b.seed), L.c_u64(env_b._ctr), L.ptr(obs_b), L.ptr(done_b), L.stream())
    for x, y in zip(ring_a, ring_b):
        assert torch.equal(x, y)
    assert torch.equal(obs_a, obs_b) and torch.equal(env_a._state, env_b._state) and bool(done_b.all())


@pytest.mark.parametrize('n,sigma,cache', [(300, 0.0, False), (4096, 0.3, True), (17, 0.3, False)])
def test_worker_step_equals_policy_action_plus_env_step_store_reset(n, sigma, cache):
    """mpg_worker_step (worker.py:95-112 as ONE launch: the policy pass of a 16-agent group, then env.step -> ring -> env.reset of those
    agents on the group's first wave) == mpg_policy_action + mpg_env_step_store_reset, bit for bit: actions (with exploration noise),
    ring rows (wrapping), env state, next observations, done flags - ragged agent counts, with and without the packed weight cache."""
    import ctypes
    import mpg_amd._lib as L
    from mpg_amd import ops
    from mpg_amd.envs import PathTrackingEnv
    from tests.golden_inputs import mlp_weights_flat
    rng = np.random.Generator(np.random.PCG64(n))
    cap, nxt = 3 * n + 5, 2 * n + 9                   # next + n wraps around the ring
    pol = torch.as_tensor(mlp_weights_flat(rng, 6, 4)).cuda()
    cfg = ops.make_cfg()
    wc = None
    if cache:                                         # the packed-image instantiation: cfg.wcache[0] -> the policy's images
        wc = ops.WeightCache(pol, [(6, 4)])
        cfg.wcache[0] = wc.pointer
    env_a, env_b = PathTrackingEnv(num_agent=n, seed=4), PathTrackingEnv(num_agent=n, seed=4)
    obs_a, obs_b = env_a.reset().clone(), env_b.reset().clone()
    ring_a = [torch.zeros(cap, 6).cuda(), torch.zeros(cap, 2).cuda(), torch.zeros(cap).cuda(), torch.zeros(cap, 6).cuda(),
              torch.zeros(cap, dtype=torch.uint8).cuda()]
    ring_b = [torch.zeros_like(t) for t in ring_a]
    done_a, done_b = torch.empty(n, dtype=torch.uint8).cuda(), torch.empty(n, dtype=torch.uint8).cuda()
    # two calls
    act_a = ops.policy_action(cfg, pol, obs_a, explore_sigma=sigma, seed=11, ctr=5)
    L.call('mpg_env_step_store_reset', L.c_int(0), L.c_int(n), L.c_int(6), L.ptr(env_a._state), L.ptr(act_a), L.c_int(cap), L.c_int(nxt),
           *[L.ptr(t) for t in ring_a], L.c_u64(env_a.seed), L.c_u64(env_a._ctr), L.ptr(obs_a), L.ptr(done_a), L.stream())
    # one call
    act_b = torch.empty(n, 2).cuda()
    L.call('mpg_worker_step', ctypes.byref(cfg), L.ptr(pol), L.c_int(n), L.ptr(env_b._state), L.ptr(obs_b), L.c_float(sigma), L.c_u64(11),
           L.c_u64(5), L.ptr(act_b), L.c_int(cap), L.c_int(nxt), *[L.ptr(t) for t in ring_b], L.c_u64(env_b.seed), L.c_u64(env_b._ctr),
           L.ptr(done_b), L.ptr(None), L.c_int(0), L.ptr(None), L.ptr(None), L.ptr(None), L.ptr(None), L.stream())
    torch.cuda.synchronize()
    assert torch.equal(act_a, act_b)
    for x, y in zip(ring_a, ring_b):
        assert torch.equal(x, y)
    assert torch.equal(obs_a, obs_b) and torch.equal(env_a._state, env_b._state) and torch.equal(done_a, done_b)
    del wc


def test_pre_gathered_draw_equals_the_draw_inside_the_gradient_launch():
    """mpg_env_step_store_reset_draw + mpg_mpg_gradients(draw.pre_gathered = 1) == mpg_env_step_store_reset +
    mpg_mpg_gradients(draw): same ring, same minibatch (indices, five columns), same targets and gradients, bit for bit -
    with a full ring whose fresh window wraps around the end, so that some drawn rows fall into the slots the env launch
    is writing (those are left to the gradient launch)."""
    import ctypes
    import mpg_amd._lib as L
    from mpg_amd import ops
    from mpg_amd.envs import PathTrackingEnv
    from tests.golden_inputs import mlp_weights_flat

    class Draw(ctypes.Structure):               # mpg_replay_draw_t, include/mpg_hip.h
        _fields_ = [('n_storage', ctypes.c_int), ('seed', ctypes.c_uint64), ('ctr', ctypes.c_uint64),
                    ('ring_obs', ctypes.c_void_p), ('ring_act', ctypes.c_void_p), ('ring_rew', ctypes.c_void_p),
                    ('ring_obs2', ctypes.c_void_p), ('ring_done', ctypes.c_void_p), ('idx_out', ctypes.c_void_p),
                    ('done_out', ctypes.c_void_p), ('pre_gathered', ctypes.c_int), ('capacity', ctypes.c_int),
                    ('fresh_start', ctypes.c_int), ('fresh_count', ctypes.c_int)]

    def dev(x):
        return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).cuda()

    n, cap, nxt, B = 64, 1000, 970, 512               # 970 + 64 wraps; ~6 % of the draws land in the fresh window
    rng = np.random.Generator(np.random.PCG64(11))
    params = dev(np.concatenate([mlp_weights_flat(rng, 8, 1), mlp_weights_flat(rng, 8, 1), mlp_weights_flat(rng, 6, 4)]))
    targets = (params * 0.97).contiguous()
    act = dev(rng.uniform(-1.2, 1.2, (n, 2)))
    ring0 = [dev(rng.standard_normal((cap, 6)) * np.array([3, 1, .5, 1, .5, 300])), dev(rng.uniform(-1, 1, (cap, 2))),
             dev(rng.standard_normal(cap)), dev(rng.standard_normal((cap, 6)) * np.array([3, 1, .5, 1, .5, 300])),
             torch.as_tensor(rng.integers(0, 2, cap).astype(np.uint8)).cuda()]
    cfg = ops.make_cfg()
    nb = L.lib().mpg_mpg_gradients_workspace_bytes(ctypes.byref(cfg), L.c_int(B), L.c_int(1), L.c_int(25), L.c_int(2), L.c_int(2))
    sel, w = (ctypes.c_int * 2)(0, 25), (ctypes.c_float * 2)(0.3, 0.7)

    def run(pre):
        env = PathTrackingEnv(num_agent=n, seed=5)
        env.reset()
        ring = [t.clone() for t in ring0]
        out = dict(obs=torch.zeros(B, 6).cuda(), act=torch.zeros(B, 2).cuda(), rew=torch.zeros(B).cuda(), obs2=torch.zeros(B, 6).cuda(),
                   idx=torch.zeros(B, dtype=torch.int32).cuda(), done=torch.zeros(B).cuda(), y=torch.zeros(B).cuda(),
                   grad=torch.zeros(params.numel()).cuda(), stats=torch.zeros(16).cuda(), wobs=torch.empty(n, 6).cuda())
        d = Draw(cap, 77, 5, *[t.data_ptr() for t in ring], out['idx'].data_ptr(), out['done'].data_ptr(), 0, 0, 0, 0)
        env_args = [L.c_int(0), L.c_int(n), L.c_int(6), L.ptr(env._state), L.ptr(act), L.c_int(cap), L.c_int(nxt),
                    *[L.ptr(t) for t in ring], L.c_u64(env.seed), L.c_u64(env._ctr), L.ptr(out['wobs']), L.ptr(None)]
        if pre:
            L.call('mpg_env_step_store_reset_draw', *env_args, ctypes.byref(d), L.c_int(B), L.ptr(out['obs']), L.ptr(out['act']),
                   L.ptr(out['rew']), L.ptr(out['obs2']), L.stream())
            d.pre_gathered, d.capacity, d.fresh_start, d.fresh_count = 1, cap, nxt, n
        else:
            L.call('mpg_env_step_store_reset', *env_args, L.stream())
        ws = torch.empty(nb + 256, dtype=torch.uint8, device='cuda')
        L.call('mpg_mpg_gradients', ctypes.byref(cfg), L.c_int(2), L.ptr(params), L.ptr(targets), L.c_int(B), L.ptr(out['obs']),
               L.ptr(out['act']), L.ptr(out['rew']), L.ptr(out['obs2']), L.ptr(None), L.c_int(1), L.c_int(25), sel, L.c_int(2), w,
               L.ptr(None), L.c_u64(3), L.c_u64(1), L.c_float(1.0 / B), L.ptr(out['grad']), L.ptr(out['stats']), L.ptr(out['y']),
               L.ptr(None), ctypes.byref(d), L.ptr(ws), L.c_size_t(ws.numel()), L.stream())
        torch.cuda.synchronize()
        return out, ring
    a, ring_a = run(False)
    b, ring_b = run(True)
    for x, y in zip(ring_a, ring_b):
        assert torch.equal(x, y)
    idx = a['idx'].long()
    fresh = ((idx - nxt) % cap) < n
    assert 5 < int(fresh.sum()) < 100                 # the window is exercised
    assert torch.equal(a['obs'], ring_a[0][idx]) and torch.equal(a['obs2'], ring_a[3][idx])     # and it is the post-add ring
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_future_data_observations_vs_reference_run(golden):
    """num_future_data = 3 (path_tracking_env.py:385-402): six base entries + three look-ahead delta-y terms, against the
    reference's own env on the same start states and actions; and the reset() branch against the oracle's statement of
    _get_obs applied to the kernel's own drawn state."""
    from oracle import mpg_oracle as O
    from mpg_amd.envs import PathTrackingEnv
    g = golden('env_future_ref.npz')
    K, N = int(g['K']), g['obs0'].shape[0]
    env = PathTrackingEnv(num_future_data=K, num_agent=N)
    assert env.obs_dim == 6 + K
    env.reset(init_obs=torch.as_tensor(np.concatenate([g['obs0'], np.zeros((N, K), np.float32)], 1)).cuda())
    for t in range(g['actions'].shape[0]):
        o, r, d, _ = env.step(torch.as_tensor(g['actions'][t]).cuda())
        assert o.shape == (N, 6 + K)
        ref = g['obs'][t]
        tol = 5e-6 * (1.0 + np.abs(ref))            # the stated env bar: 5e-6 abs + ulp-of-x terms (x up to 1200)
        assert (np.abs(o.cpu().numpy() - ref) <= tol + 6e-5 * (np.arange(6 + K) == 5)).all(), t
        np.testing.assert_allclose(r.cpu().numpy(), g['reward'][t], rtol=2e-6, atol=1e-5)
    for n in (1, 63, 4099):                          # ragged sizes, reset() branch, an odd obs stride (7)
        env = PathTrackingEnv(num_future_data=1, num_agent=n, seed=5)
        obs = env.reset().cpu().numpy()
        fs = env.veh_full_state.cpu().numpy()
        vs = env.veh_state.cpu().numpy()
        ref = O.PathTrackingEnvOracle(n, num_future_data=1)._get_obs(vs, fs)
        np.testing.assert_allclose(obs, ref, rtol=0, atol=2e-5)
        plain = PathTrackingEnv(num_future_data=0, num_agent=n, seed=5)
        assert np.array_equal(plain.reset().cpu().numpy(), obs[:, :6])      # the base entries do not depend on K


def test_cart_pole_env_vs_float64_restatement():
    """InvertedPendulumContiEnv (analytic RK4 statement of inverted_pendulum_conti.xml; PARITY UNPINNED - no MuJoCo).
    The float32 kernel against the float64 restatement of the same equations: 50 steps (2 s) with random actions, each
    step checked from the kernel's own previous state (<= 3e-6 relative-to-scale per step); reward / done / clipping /
    reset range / fused step-store-reset."""
    import mpg_amd._lib as L
    from oracle import mpg_oracle as O
    from mpg_amd.envs import InvertedPendulumContiEnv
    rng = np.random.Generator(np.random.PCG64(11))
    for n in (1, 64, 1000):
        env = InvertedPendulumContiEnv(num_agent=n, seed=3)
        obs0 = env.reset().cpu().numpy().astype(np.float64)
        assert obs0.shape == (n, 4) and (np.abs(obs0) <= 0.01).all() and np.abs(obs0).max() > 0.005 * (n > 1)
        ref = O.InvertedPendulumContiOracle(n)
        prev = obs0
        for t in range(50):
            a = rng.uniform(-3.5, 3.5, (n, 1)).astype(np.float32)           # beyond ctrlrange: clipped inside
            o, r, d, _ = env.step(torch.as_tensor(a).cuda())
            # one-step error from the kernel's own previous state (open loop, a swinging pole separates two float
            # precisions exponentially - that would measure the pendulum, not the kernel)
            ref.reset(init_obs=prev)
            ro, rr, rd, _ = ref.step(a[:, 0].astype(np.float64))
            prev = o.cpu().numpy().astype(np.float64)
            scale = 1.0 + np.abs(ro)
            assert (np.abs(prev - ro) / scale).max() <= 3e-6, (n, t)
            np.testing.assert_allclose(r.cpu().numpy(), rr, rtol=2e-5, atol=1e-5)
            # done may legitimately differ for an agent sitting within rounding of a threshold
            near = (np.abs(np.abs(ro[:, 0]) - 2.0) < 1e-4) | (np.abs(np.abs(ro[:, 1]) - 0.2) < 1e-4)
            assert (d.cpu().numpy().astype(bool) == rd)[~near].all()
        assert bool(env.done.any()) or n == 1        # random +-3 pushes for 2 s drop the pole for essentially every agent
    # reset(init_obs) + reset() of the done agents only
    env = InvertedPendulumContiEnv(num_agent=8, seed=1)
    init = torch.as_tensor(np.array([[0., 0., 0., 0.]] * 4 + [[2.5, 0., 0., 0.]] * 4, np.float32)).cuda()
    env.reset(init_obs=init)
    o, r, d, _ = env.step(torch.zeros(8, 1).cuda())
    assert d.cpu().tolist() == [0, 0, 0, 0, 1, 1, 1, 1]
    o2 = env.reset().cpu().numpy()
    assert np.array_equal(o2[:4], o.cpu().numpy()[:4]) and (np.abs(o2[4:]) <= 0.01).all()
    # fused step + ring store + reset == the separate calls, bit for bit
    n, cap, nxt = 50, 64, 40
    act = torch.as_tensor(rng.uniform(-3, 3, (n, 1)).astype(np.float32)).cuda()
    ea, eb = InvertedPendulumContiEnv(num_agent=n, seed=9), InvertedPendulumContiEnv(num_agent=n, seed=9)
    start = torch.as_tensor((rng.standard_normal((n, 4)) * np.array([1.5, 0.15, 0.5, 0.5])).astype(np.float32)).cuda()
    ea.reset(init_obs=start.clone())
    eb.reset(init_obs=start.clone())
    o2, r, d, _ = ea.step(act)
    ring_a = [torch.zeros(cap, 4).cuda(), torch.zeros(cap, 1).cuda(), torch.zeros(cap).cuda(), torch.zeros(cap, 4).cuda(),
              torch.zeros(cap, dtype=torch.uint8).cuda()]
    L.call('mpg_replay_add', L.c_int(cap), L.c_int(nxt), L.c_int(n), L.c_int(4), L.c_int(1), L.ptr(start), L.ptr(act), L.ptr(r),
           L.ptr(o2), L.ptr(d), *[L.ptr(t) for t in ring_a], L.stream())
    obs_a = ea.reset()
    ring_b = [torch.zeros_like(t) for t in ring_a]
    obs_b, done_b = torch.empty(n, 4).cuda(), torch.empty(n, dtype=torch.uint8).cuda()
    L.call('mpg_env_step_store_reset', L.c_int(1), L.c_int(n), L.c_int(4), L.ptr(eb._state), L.ptr(act), L.c_int(cap), L.c_int(nxt),
           *[L.ptr(t) for t in ring_b], L.c_u64(eb.seed), L.c_u64(eb._ctr), L.ptr(obs_b), L.ptr(done_b), L.stream())
    for x, y in zip(ring_a, ring_b):
        assert torch.equal(x, y)
    assert torch.equal(obs_a, obs_b) and torch.equal(ea._state[:4], eb._state[:4]) and torch.equal(done_b, d)
    assert 0 < int(d.sum().item()) < n
