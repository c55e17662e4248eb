"""HIP network kernels (forward / critic targets / critic loss+gradient) through the C ABI against the oracle
and the reference-generated goldens.

Tolerances (float32 path, exact-fp32 MFMA): <= 1e-4 relative L2 per gradient array, forward
values <= 2e-5 relative-to-scale; float64 oracle used as the yard-stick where the fixture holds one."""
import numpy as np
import pytest
import torch

from oracle import mpg_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).to(DEV)


def rand_net(rng, din, dout):
    ws = [rng.standard_normal((din, 256)) * 0.4, rng.standard_normal(256) * 0.1,
          rng.standard_normal((256, 256)) * (1.4 / 16), rng.standard_normal(256) * 0.1,
          rng.standard_normal((256, dout)) * 0.1, rng.standard_normal(dout) * 0.1]
    return np.concatenate([w.ravel() for w in ws]).astype(np.float32)


@pytest.mark.parametrize('rows', [1, 15, 16, 17, 255, 4096, 4099])
@pytest.mark.parametrize('shape', [(6, 4, 2, 1), (8, 1, 1, 0), (4, 2, 1, 0), (5, 1, 1, 0)])
def test_mlp_forward_vs_oracle(rows, shape):
    from mpg_amd import ops
    din, dout, used, act = shape
    rng = np.random.Generator(np.random.PCG64(rows * 31 + din))
    flat = rand_net(rng, din, dout)
    x = rng.standard_normal((rows, din)).astype(np.float32)
    scale = rng.uniform(0.5, 2.0, din)
    scale[16:] = 1.0                      # the scale vector has 16 entries (mpg_cfg_t.obs_scale: the observation part of an input)
    y = ops.mlp_forward(dev(flat), din, dout, used, act, dev(x), in_scale=scale[:16], n_scaled=min(din, 16)).cpu().numpy()
    ws = O.unflatten(flat, din, 256, dout, dtype=torch.float64)
    ref = O.mlp(ws, torch.as_tensor(x, dtype=torch.float64) * torch.as_tensor(scale.astype(np.float32)).double(),
                'tanh' if act else 'linear').numpy()[:, :used]
    assert np.abs(y - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize('rows', [1, 17, 4099])
@pytest.mark.parametrize('shape', [(7, 4, 2, 1), (9, 4, 2, 1), (14, 4, 2, 1), (16, 4, 2, 1), (9, 1, 1, 0), (11, 1, 1, 0), (16, 1, 1, 0), (17, 1, 1, 0), (18, 1, 1, 0),
                                   (24, 1, 1, 0)])
def test_mlp_forward_wide_first_layer_vs_oracle(rows, shape):
    """first layers 7 .. 24 wide (observations with look-ahead entries, num_future_data <= 10: policy 6 + K, critics 8 + K): the 16- and
    24-column instantiations"""
    test_mlp_forward_vs_oracle(rows, shape)


def _cfg_nets(g, names, env='PathTracking-v0'):
    from mpg_amd import ops
    cfg = ops.make_cfg(env)
    online = {k: g['w_' + k] for k in names}
    targets = {k: (g['w_' + k] * np.float32(g['target_scale'])).astype(np.float32) for k in names}
    return cfg, online, targets


def test_policy_action_and_clipped_double_q_target_vs_golden(golden):
    from mpg_amd import ops
    g = golden('mpg_v2_H256_B64.npz')
    cfg, online, targets = _cfg_nets(g, ['Q1', 'Q2', 'policy'])
    y = ops.q_targets(cfg, dev(targets['policy']), dev(targets['Q1']), dev(targets['Q2']),
                      dev(g['batch_rewards']), dev(g['batch_obs_tp1'])).cpu().numpy()
    np.testing.assert_allclose(y, g['it100_targets'], rtol=2e-5, atol=2e-6)
    # deterministic action = oracle action
    ocfg = O.Cfg()
    nets = O.Nets(ocfg, online, target_scale=g['target_scale'], dtype=torch.float64)
    a_ref = nets.compute_action(O.process_obses(ocfg, torch.as_tensor(g['batch_obs']).double())).detach().numpy()
    a = ops.policy_action(cfg, dev(online['policy']), dev(g['batch_obs'])).cpu().numpy()
    np.testing.assert_allclose(a, a_ref, rtol=0, atol=2e-6)


def test_target_kernel_two_groups_per_workgroup_equals_one():
    """From 256 row groups on, k_target_fused sends two groups through every network per workgroup (forward_group2).  A row's
    target does not depend on its neighbours: 4107 rows (257 groups, ragged last one, odd count) in one call equal the same
    rows in two calls of 128 + 129 groups, which take the one-group path - bit for bit, with and without the smoothing noise."""
    from mpg_amd import ops
    rng = np.random.Generator(np.random.PCG64(21))
    cfg = ops.make_cfg()
    pol, q1, q2 = dev(rand_net(rng, 6, 4)), dev(rand_net(rng, 8, 1)), dev(rand_net(rng, 8, 1))
    n = 4107
    obs2 = dev(rng.standard_normal((n, 6)) * np.array([3, 1, .5, 1, .5, 300]))
    rew = dev(rng.standard_normal(n))
    eps = dev(rng.standard_normal((n, 2)))
    for sm in (None, eps):
        cut = 2048
        whole = ops.q_targets(cfg, pol, q1, q2, rew, obs2, smooth_eps=sm).clone()
        a = ops.q_targets(cfg, pol, q1, q2, rew[:cut].contiguous(), obs2[:cut].contiguous(),
                          smooth_eps=None if sm is None else sm[:cut].contiguous()).clone()
        b = ops.q_targets(cfg, pol, q1, q2, rew[cut:].contiguous(), obs2[cut:].contiguous(),
                          smooth_eps=None if sm is None else sm[cut:].contiguous()).clone()
        assert torch.equal(whole, torch.cat([a, b]))


def test_td3_smoothed_target_vs_golden(golden):
    from mpg_amd import ops
    g = golden('td3_H256_B64.npz')
    cfg, online, targets = _cfg_nets(g, ['Q1', 'Q2', 'policy'])
    y = ops.q_targets(cfg, dev(targets['policy']), dev(targets['Q1']), dev(targets['Q2']), dev(g['batch_rewards']),
                      dev(g['batch_obs_tp1']), smooth_eps=dev(g['smooth_eps']), smooth_sigma=0.2,
                      smooth_clip=0.5).cpu().numpy()
    np.testing.assert_allclose(y, g['targets'], rtol=2e-5, atol=2e-6)


def test_exploration_noise_statistics():
    from mpg_amd import ops
    cfg = ops.make_cfg()
    rng = np.random.Generator(np.random.PCG64(3))
    flat = rand_net(rng, 6, 4)
    obs = dev(np.zeros((1 << 15, 6), np.float32))
    a0 = ops.policy_action(cfg, dev(flat), obs)
    a1 = ops.policy_action(cfg, dev(flat), obs, explore_sigma=0.1, seed=5, ctr=0)
    a2 = ops.policy_action(cfg, dev(flat), obs, explore_sigma=0.1, seed=5, ctr=1)
    d = (a1 - a0).cpu().numpy()
    assert abs(d.mean()) < 3e-3 and abs(d.std() - 0.1) < 3e-3
    assert np.abs(np.corrcoef(d[:, 0], d[:, 1])[0, 1]) < 0.03
    assert (a1 - a2).abs().max().item() > 0.05          # a new counter gives a new stream
    a1b = ops.policy_action(cfg, dev(flat), obs, explore_sigma=0.1, seed=5, ctr=0)
    assert torch.equal(a1, a1b)                          # counter-based: reproducible


def _unclip(flat_clipped, norm, clip=3.0):
    scale = clip * min(1.0 / float(norm), 1.0 / clip)
    return flat_clipped / scale


@pytest.mark.parametrize('fixture,loss_keys', [('mpg_v2_H256_B64.npz', ('it100_q_loss1', 'it100_q_loss2')),
                                               ('td3_H256_B64.npz', ('q_loss1', 'q_loss2'))])
def test_q_loss_grad_vs_golden(golden, fixture, loss_keys):
    from mpg_amd import ops
    g = golden(fixture)
    cfg, online, _ = _cfg_nets(g, ['Q1', 'Q2', 'policy'])
    pre = 'it100_' if 'mpg' in fixture else ''
    y = dev(g[pre + 'targets'])
    nq = ops.q_size(cfg)
    for i, nm in enumerate(('Q1', 'Q2')):
        loss, grad, td = ops.q_loss_grad(cfg, dev(online[nm]), dev(g['batch_obs']), dev(g['batch_actions']), y,
                                         want_td=True)
        np.testing.assert_allclose(loss.item(), g[loss_keys[i]], rtol=5e-5)
        norm = g[pre + 'q_gradient_norm%d' % (i + 1)]
        ref = _unclip(g[pre + 'grads'][i * nq:(i + 1) * nq], norm)
        got = grad.cpu().numpy()
        assert abs(np.linalg.norm(got) - norm) <= 1e-4 * norm
        # per-array check (W1,b1,W2,b2,W3,b3)
        o = 0
        for shp in O.mlp_shapes(8, 256, 1):
            n = int(np.prod(shp))
            assert rel_l2(got[o:o + n], ref[o:o + n]) <= 1e-4, (nm, shp)
            o += n


@pytest.mark.parametrize('rows,K', [(1, 0), (17, 0), (300, 0), (4096, 0), (17, 1), (300, 3), (4096, 8), (300, 9), (4096, 10)])
def test_q_loss_grad_vs_oracle_autograd_ragged(rows, K):
    """K look-ahead entries (num_future_data): critic input 8 + K wide"""
    from mpg_amd import ops
    rng = np.random.Generator(np.random.PCG64(rows))
    scale = list(O.OBS_SCALE_PT) + [1.] * K
    cfg = ops.make_cfg(obs_dim=6 + K, obs_scale=scale)
    flat = rand_net(rng, 8 + K, 1)
    obs = (rng.standard_normal((rows, 6 + K)) * np.array([3, 1, .5, 1, .5, 300] + [1.] * K)).astype(np.float32)
    act = rng.uniform(-1, 1, (rows, 2)).astype(np.float32)
    y = rng.standard_normal(rows).astype(np.float32)
    loss, grad, td = ops.q_loss_grad(cfg, dev(flat), dev(obs), dev(act), dev(y), want_td=True)
    ocfg = O.Cfg(obs_dim=6 + K, obs_scale=scale)
    ws = O.unflatten(flat, 8 + K, 256, 1, dtype=torch.float64, requires_grad=True)
    po = O.process_obses(ocfg, torch.as_tensor(obs).double())
    q = O.mlp(ws, torch.cat([po, torch.as_tensor(act).double()], 1), 'linear')[:, 0]
    l = 0.5 * torch.mean((q - torch.as_tensor(y).double()) ** 2)
    gs = torch.autograd.grad(l, ws)
    ref = np.concatenate([x.numpy().ravel() for x in gs])
    np.testing.assert_allclose(loss.item(), l.item(), rtol=2e-5)
    np.testing.assert_allclose(td.cpu().numpy(), (q.detach().numpy() - y), rtol=0, atol=3e-5)
    got = grad.cpu().numpy()
    o = 0
    assert got.size == ref.size
    for shp in O.mlp_shapes(8 + K, 256, 1):
        n = int(np.prod(shp))
        assert rel_l2(got[o:o + n], ref[o:o + n]) <= 2e-5, (rows, shp, rel_l2(got[o:o + n], ref[o:o + n]))
        o += n


def test_nstep_target_vs_golden(golden):
    """MPG-v1: real-env 25-step rollout on the HIP env + n-step return (mpg_learner.py:146-169)."""
    from mpg_amd import ops
    from mpg_amd.envs import PathTrackingEnv
    g = golden('mpg_v1_H256_B64.npz')
    cfg, online, targets = _cfg_nets(g, ['Q1', 'policy'])
    B = g['batch_obs'].shape[0]
    env = PathTrackingEnv(num_agent=B)
    obs = dev(g['batch_obs'])
    env.reset(init_obs=obs)
    rewards = []
    for t in range(25):
        a = dev(g['batch_actions']) if t == 0 else ops.policy_action(cfg, dev(online['policy']), obs)
        obs, r, _, _ = env.step(a)
        rewards.append(r)
    rewards = torch.stack(rewards).contiguous()
    # 25 closed-loop env steps: measured max relative reward error 3.7e-6 at step 24 (tools/v1_errors.py); the spread
    # between two valid float32 evaluations of the REFERENCE itself (library vs correctly rounded sin/cos/atan) is 3.2e-6
    np.testing.assert_allclose(rewards.cpu().numpy(), g['nstep_all_rewards'], rtol=2e-5, atol=2e-6)
    y = ops.nstep_targets(cfg, dev(targets['policy']), dev(targets['Q1']), rewards, obs).cpu().numpy()
    from tests import yardstick as Y
    Y.check_values(y, g['it100_targets'], g['it100_targets_f64'], what='n-step targets')


@pytest.mark.parametrize('shapes', [[(8, 1), (8, 1), (6, 4)], [(5, 1), (4, 2)]])
def test_optimizer_launch_keeps_both_packed_images_current(shapes):
    """mpg_clip_adam_polyak rewrites the packed register images of the hidden kernels it updates (parameters AND targets, forward and
    transposed image each).  After an update both images of both buffers must equal a fresh mpg_weight_cache_pack of the updated
    buffers, bit for bit, and the update itself must equal the launch without cache descriptors."""
    from mpg_amd import ops
    from tests.golden_inputs import mlp_weights_flat
    rng = np.random.Generator(np.random.PCG64(23))
    sizes = [ops.net_size(i, o) for i, o in shapes]
    n, k = sum(sizes), len(shapes)
    flat = dev(np.concatenate([mlp_weights_flat(rng, i, o) for i, o in shapes]))
    tgt = (flat * 0.9).contiguous()
    m, v = dev(rng.standard_normal(n) * 0.01), dev(rng.random(n) * 1e-3)
    g = dev(rng.standard_normal(n) * 0.02)
    wc, wct = ops.WeightCache(flat, shapes), ops.WeightCache(tgt, shapes)
    plain = [t.clone() for t in (flat, m, v, tgt, g)]
    norms, norms2 = torch.zeros(k, device=DEV), torch.zeros(k, device=DEV)
    lr, da, dp = [1e-2] * k, [1] * k, [1] * (k - 1) + [0]
    ops.clip_adam_polyak(flat, m, v, tgt, g, ops.sq_partials(g, sizes), sizes, 3.0, lr, da, dp, 0.005, norms, wc_w=wc, wc_target=wct)
    ops.clip_adam_polyak(plain[0], plain[1], plain[2], plain[3], plain[4], ops.sq_partials(plain[4], sizes), sizes, 3.0, lr, da, dp, 0.005, norms2)
    for x, y in zip((flat, m, v, tgt, g, norms), plain + [norms2]):
        assert torch.equal(x, y)
    for c in (wc, wct):
        got = c.packed.clone()
        c.packed.zero_()
        c.pack()
        assert torch.equal(got, c.packed)


def test_weight_cache_is_bit_identical_to_the_strided_path():
    """mpg_wcache_t: packed register images are an acceleration only - forward, backward and rollout results are
    bit-identical with and without a descriptor in the cfg, and stay so after mpg_adam_polyak (which is handed the
    descriptor) rewrites the buffer.  The descriptor is a caller-owned object: a cfg without it never sees the images."""
    from mpg_amd import ops
    from tests.golden_inputs import mlp_weights_flat, reset_law_obs
    rng = np.random.Generator(np.random.PCG64(17))
    plain, cached = ops.make_cfg(), ops.make_cfg()
    nq, npol = ops.q_size(plain), ops.policy_size(plain)
    flat = dev(np.concatenate([mlp_weights_flat(rng, 8, 1), mlp_weights_flat(rng, 6, 4)]))
    q1, pol = flat[:nq], flat[nq:]
    B = 100
    obs, act = dev(reset_law_obs(rng, B)), dev(rng.uniform(-1, 1, (B, 2)))
    y = dev(rng.standard_normal(B))
    eps = dev(rng.standard_normal((25, B)))

    def run(cfg):
        a = ops.policy_action(cfg, pol, obs)
        _, g, _ = ops.q_loss_grad(cfg, q1, obs, act, y)
        _, _, pg = ops.rollout_pg(cfg, pol, q1, obs, eps, [0, 25], [0.5, 0.5])
        return a.clone(), g.clone(), pg.clone()
    ref = run(plain)
    wc = ops.WeightCache(flat, [(8, 1), (6, 4)])
    cached.wcache[0] = wc.pointer
    got = run(cached)
    for r, g in zip(ref, got):
        assert torch.equal(r, g)
    # an Adam step through the ABI keeps the packed images current by itself when it is given the descriptor
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    grad = dev(rng.standard_normal(nq + npol) * 0.01)
    ops.adam_polyak(flat, m, v, None, grad, [nq, npol], [1e-2, 1e-2], [1, 1], [0, 0], 0.005, wc_w=wc)
    got2 = run(cached)
    ref2 = run(plain)                              # strided loads from the updated weights
    for r, g in zip(ref2, got2):
        assert torch.equal(r, g)
    assert not torch.equal(ref[0], ref2[0])
    # mpg_mlp_forward takes the descriptor as an explicit argument
    x = torch.cat([obs, act], 1).contiguous()
    qa = ops.mlp_forward(q1, 8, 1, 1, ops.ACT_LINEAR, x)
    qb = ops.mlp_forward(q1, 8, 1, 1, ops.ACT_LINEAR, x, wcache=wc)
    assert torch.equal(qa, qb)
    # a write that by-passes the ABI leaves the images stale until the owner re-packs them
    flat.mul_(1.5)
    stale = ops.policy_action(cached, pol, obs)
    wc.pack()
    fresh = ops.policy_action(cached, pol, obs)
    assert torch.equal(fresh, ops.policy_action(plain, pol, obs)) and not torch.equal(stale, fresh)


def test_tanh_policy_with_action_range_is_refused():
    """policy_out_activation='tanh' together with an action_range would be range*tanh(tanh(z)) in the reference
    (policy.py:176-177,197-199); the kernels only implement z / tanh(z) / range*tanh(z), so the ABI rejects the
    combination instead of silently computing something else."""
    import mpg_amd._lib as L
    from mpg_amd import ops
    from tests.golden_inputs import mlp_weights_flat
    rng = np.random.Generator(np.random.PCG64(3))
    cfg = ops.make_cfg('PathTracking-v0', policy_out_activation='tanh', action_range=2.0)
    with pytest.raises(L.MpgError):
        ops.policy_action(cfg, dev(mlp_weights_flat(rng, 6, 4)), dev(rng.standard_normal((16, 6))))

