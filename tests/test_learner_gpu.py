"""End-to-end: the device learner / worker / buffer / optimizer classes (reference class and method names) against
the goldens produced by the reference's own MPGLearner.compute_gradient, plus a short training run."""
import numpy as np
import pytest
import torch

from oracle import mpg_oracle as O
from tests import yardstick as Y

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).to(DEV)


def _learner(g, version, B=64, K=0):
    from mpg_amd.config import default_args
    from mpg_amd.learners import MPGLearner
    from mpg_amd.policy import PolicyWithQs
    args = default_args('MPG-' + version, replay_batch_size=B, num_batch_reuse=1, num_future_data=K)
    learner = MPGLearner(PolicyWithQs, args)
    pw = learner.policy_with_value
    flat = np.concatenate([g['w_' + n] for n in pw.names])
    pw.set_flat(flat, (flat * np.float32(g['target_scale'])).astype(np.float32))
    return learner


@pytest.mark.parametrize('version,K', [('v2', 0), ('v1', 0), ('v2', 3), ('v2', 10)])
def test_compute_gradient_vs_reference_golden(golden, version, K):
    """The full list the reference's MPGLearner.compute_gradient returns (clipped q1, (q2), policy gradients) and its
    stats, on the same minibatch, weights and model noise.  <= 1e-4 relative L2 per array.
    K = num_future_data (train_script.py:90,146-147): observations with K look-ahead entries, first layers 6+K / 8+K
    wide - the 16-column instantiations of the network kernels and of the two rollout sweeps."""
    g = golden('mpg_%s_H256_B64%s.npz' % (version, '_K%d' % K if K else ''))
    learner = _learner(g, version, K=K)
    pw = learner.policy_with_value
    assert pw.obs_dim == 6 + K and pw.cfg.obs_dim == 6 + K
    batch = [dev(g[k]) for k in ('batch_obs', 'batch_actions', 'batch_rewards', 'batch_obs_tp1', 'batch_dones')]
    for it in (100, 9000):
        learner.counter = 0
        grads = learner.compute_gradient(batch, None, None, it, eps=dev(g['eps']))
        assert len(grads) == (18 if version == 'v2' else 12)
        got = torch.cat([x.reshape(-1) for x in grads]).cpu().numpy()
        p = 'it%d_' % it
        # every array: <= 1e-4 rel-L2 vs the reference's float32 result AND error vs the reference's float64 run at most
        # 4x the reference's own float32 error (tests/yardstick.py) - the same rule for MPG-v1, whose target comes from
        # 25 real-env steps (measured: 3.6e-7 on the critic arrays, 2.3e-6 on the policy's; tools/v1_errors.py)
        Y.check_gradients(got, g[p + 'grads'], g[p + 'grads_f64'], [(n,) + tuple(pw.dims[n]) for n in pw.names],
                          where='MPG-%s it %d' % (version, it))
        st = learner.get_stats()
        for k in ('value_mean', 'policy_total_loss', 'policy_gradient_norm', 'q_loss1', 'q_gradient_norm1', 'q_loss2',
                  'q_gradient_norm2'):
            if p + k in g:
                np.testing.assert_allclose(st[k], g[p + k], rtol=2e-5, atol=1e-6, err_msg=k)
        np.testing.assert_allclose(st['w_list'], g[p + 'w_list'], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(st['all_losses'], g[p + 'all_losses'], rtol=2e-5, atol=1e-6)
        Y.check_values(learner.batch_data['batch_targets'].cpu().numpy(), g[p + 'targets'], g[p + 'targets_f64'],
                       what='targets MPG-%s' % version)
    np.testing.assert_allclose(learner.compute_td_error().cpu().numpy(), g['td_error'], rtol=1e-4, atol=2e-5)


def test_replay_buffer_vs_the_references_own_buffer(golden):
    """ReplayBuffer on the device against the reference's buffer.py (imported as-is by make_golden.py): the same add_batch
    sequence wraps the 50-slot ring more than twice; `_next_idx`, `len`, the storage content, `_encode_sample`'s five
    columns at fixed indices and `replay()`'s gate / counter are compared exactly (the reference's bool `done` column is
    a uint8 column here, and float32 in the learner-facing gather like mpg_learner.py:71 casts it)."""
    from mpg_amd.buffer import ReplayBuffer
    from mpg_amd.config import default_args
    g = golden('replay_buffer_ref.npz')
    args = default_args(max_buffer_size=int(g['capacity']), replay_starts=int(g['replay_starts']),
                        replay_batch_size=int(g['replay_batch_size']))
    rb = ReplayBuffer(args, 0)
    o = 0
    for k, m in enumerate(g['sizes']):
        sl = slice(o, o + m)
        rb.add_batch((dev(g['obs'][sl]), dev(g['act'][sl]), dev(g['rew'][sl]), dev(g['obs2'][sl]),
                      torch.as_tensor(g['done'][sl]).to(DEV)))
        o += m
        assert rb._next_idx == g['next_idx'][k] and len(rb) == g['length'][k]
        r = rb.replay()
        assert (r is not None) == bool(g['replay_gate'][k])
        if r is not None:
            assert [tuple(x.shape) for x in r] == [(16, 6), (16, 2), (16,), (16, 6), (16,), (16,)]
            idx = r[-1].long()
            assert torch.equal(r[0], rb.obs[idx]) and torch.equal(r[2], rb.rew[idx])
        enc = rb.sample_with_idxes(torch.as_tensor(g['enc%d_idx' % k]).to(DEV))
        for nm, arr in zip(('obs', 'act', 'rew', 'obs2', 'done'), enc[:5]):
            ref = g['enc%d_%s' % (k, nm)]
            assert np.array_equal(arr.cpu().numpy(), ref.astype(np.float32)) and tuple(arr.shape) == ref.shape, (k, nm)
        assert torch.equal(enc[5].cpu(), torch.as_tensor(g['enc%d_idx' % k]))
    assert rb.replay_times == int(g['replay_times'])
    n = len(rb)
    assert np.array_equal(rb.obs[:n].cpu().numpy(), g['final_obs']) and np.array_equal(rb.act[:n].cpu().numpy(), g['final_act'])
    assert np.array_equal(rb.rew[:n].cpu().numpy(), g['final_rew']) and np.array_equal(rb.obs2[:n].cpu().numpy(), g['final_obs2'])
    assert np.array_equal(rb.done[:n].cpu().numpy(), g['final_done'])


def test_evaluator_metrics_vs_reference(golden):
    """Evaluator.run_n_episodes_parallel on the HIP env + policy kernel against the reference's own
    run_n_episodes_parallel / metrics_for_an_episode (evaluator.py:118-184) from the same start states and weights: 200
    closed-loop steps, 8 episodes.  Allowance per metric: 4x the reference's own float32-vs-float64 gap (~1e-5 relative)."""
    from mpg_amd.config import default_args
    from mpg_amd.evaluator import Evaluator
    from mpg_amd.policy import PolicyWithQs
    from tests.golden_inputs import mlp_weights_flat
    g = golden('evaluator_ref.npz')
    N, T = int(g['N']), int(g['T'])
    args = default_args('MPG-v2', num_eval_agent=N, fixed_steps=T)
    ev = Evaluator(PolicyWithQs, args.env_id, args)
    pw = ev.policy_with_value
    rng = np.random.Generator(np.random.PCG64(0))
    flat = np.concatenate([mlp_weights_flat(rng, 8, 1), mlp_weights_flat(rng, 8, 1), g['w_policy']])
    pw.set_flat(flat)
    per, mean = ev.run_n_episodes_parallel(init_obs=dev(g['init_obs']))
    for j, k in enumerate(str(x) for x in g['metric_keys']):
        ref32, ref64 = g['mean'][j], g['mean_f64'][j]
        assert abs(mean[k] - ref64) <= 4 * abs(ref32 - ref64) + 2e-6 * abs(ref64) + 1e-9, (k, mean[k], ref32, ref64)
        e32 = np.abs(g['per_episode'][:, j] - g['per_episode_f64'][:, j]).max()
        got = per[k].cpu().numpy()
        assert (np.abs(got - g['per_episode_f64'][:, j]) <= 4 * e32 + 2e-6 * np.abs(g['per_episode_f64'][:, j]) + 1e-9).all(), k


@pytest.fixture(params=['split', 'f32'])
def engine(request):
    """both builds of the library (mpg_amd/_lib.py ENGINES): the split-fp16 product and the exact-fp32 engine it is measured
    against (libmpg_hip_f32.so, -DMPG_F32_MFMA) - model.py:39-43 is plain float32; every library call of the test goes to the
    selected shared object"""
    from mpg_amd import _lib as L
    with L.engine(request.param):
        yield request.param


@pytest.mark.parametrize('name', ['c2_mpg_v2_B4096', 'c3_nadp_B8192', 'c4_td3_B65536'])
def test_bench_size_cases_vs_reference(golden, name, engine):
    """BASELINE.json configs at their FULL batch sizes against the reference itself: MPG-v2 at B = 4096 (C2, the bench
    workload), NADP on the pendulum model at B = 8192 (C3), TD3 at B = 65 536 (C4).  Inputs are seeded draws regenerated
    here (tests/golden_inputs.py); the fixture holds what the unmodified reference computed from them.  Every gradient
    array: <= 1e-4 rel-L2 and within 4x the reference's own float32 error of the float64 run."""
    from mpg_amd.config import default_args
    from mpg_amd.learners import MPGLearner, NADPLearner, TD3Learner
    from mpg_amd.policy import PolicyWithQs
    from tests.golden_inputs import bench_case_inputs
    g = golden('bench_%s.npz' % name)
    d = bench_case_inputs(name)
    B = d['B']
    flat = {k: np.concatenate([np.asarray(w).ravel() for w in v]).astype(np.float32) for k, v in d['nets'].items()}
    batch = [dev(x) for x in d['batch']]
    if d['kind'] == 'MPG-v2':
        learner = MPGLearner(PolicyWithQs, default_args('MPG-v2', replay_batch_size=B, num_batch_reuse=1))
        runs = [(it, dict(eps=dev(d['eps']))) for it in (100, 9000)]
        stat_keys = ('value_mean', 'policy_total_loss', 'policy_gradient_norm', 'q_loss1', 'q_gradient_norm1', 'q_loss2',
                     'q_gradient_norm2')
    elif d['kind'] == 'NADP':
        learner = NADPLearner(PolicyWithQs, default_args('NADP', replay_batch_size=B))
        runs = [(0, dict(eps_q=dev(d['eps_q']), eps_pi=dev(d['eps_pi'])))]
        stat_keys = ('q_loss', 'policy_loss', 'value_mean', 'q_gradient_norm', 'policy_gradient_norm')
    else:
        learner = TD3Learner(PolicyWithQs, default_args('TD3', replay_batch_size=B))
        runs = [(0, dict(smooth_eps=dev(d['smooth_eps'])))]
        stat_keys = ('q_loss1', 'q_loss2', 'policy_loss', 'value_mean', 'value_var', 'q_gradient_norm1', 'q_gradient_norm2',
                     'policy_gradient_norm')
    pw = learner.policy_with_value
    w = np.concatenate([flat[n] for n in pw.names])
    pw.set_flat(w, (w * np.float32(g['target_scale'])).astype(np.float32))
    for it, kw in runs:
        learner.counter = 0
        grads = learner.compute_gradient(batch, None, None, it, **kw)
        got = torch.cat([x.reshape(-1) for x in grads]).cpu().numpy()
        p = 'it%d_' % it
        Y.check_gradients(got, g[p + 'grads'], g[p + 'grads_f64'], [(n,) + tuple(pw.dims[n]) for n in pw.names],
                          where='%s it %d' % (name, it))
        if p + 'targets_sub' in g:
            Y.check_values(learner.batch_data['batch_targets'].cpu().numpy()[::8], g[p + 'targets_sub'],
                           g[p + 'targets_sub_f64'], what=name + ' targets')
        st = learner.get_stats()
        for k in stat_keys:
            np.testing.assert_allclose(st[k], g[p + k], rtol=5e-5, atol=1e-6, err_msg=k)
        assert int(pw.nonfinite.sum().item()) == 0


def test_trained_networks_vs_reference(golden, engine):
    """The bench workload's gradient (MPG-v2, B = 4096, iterations 100 and 9000) on TRAINED networks: the online weights after
    20 000 iterations of the HIP path at the reference's default learning rates (tools/train_export.py: evaluation return
    -4365 -> -5; tests/golden/trained_weights.npz), every output computed by the unmodified reference from them
    (make_golden.py --only trained_c2).  Same rule as for the freshly initialised nets of every other fixture: <= 1e-4 rel-L2
    and within 4x the reference's own float32 error of its float64 run.  Trained hidden kernels (max |W2| 0.4) and
    activations (max |h1| 4.2) sit three orders of magnitude inside the split-fp16 engine's envelope (1023.5 / 4094), and
    the status word stays clear."""
    from mpg_amd.config import default_args
    from mpg_amd.learners import MPGLearner
    from mpg_amd.policy import PolicyWithQs
    from tests.golden_inputs import bench_case_inputs
    g = golden('trained_c2_mpg_v2_B4096.npz')
    tw = golden('trained_weights.npz')
    d = bench_case_inputs('c2_mpg_v2_B4096')
    B = d['B']
    batch = [dev(x) for x in d['batch']]
    learner = MPGLearner(PolicyWithQs, default_args('MPG-v2', replay_batch_size=B, num_batch_reuse=1))
    pw = learner.policy_with_value
    w = np.concatenate([tw['w_' + n] for n in pw.names]).astype(np.float32)
    pw.set_flat(w, (w * np.float32(g['target_scale'])).astype(np.float32))
    worst = 0.0
    for it in (100, 9000):
        learner.counter = 0
        grads = learner.compute_gradient(batch, None, None, it, eps=dev(d['eps']))
        got = torch.cat([x.reshape(-1) for x in grads]).cpu().numpy()
        p = 'it%d_' % it
        worst = max(worst, Y.check_gradients(got, g[p + 'grads'], g[p + 'grads_f64'], [(n,) + tuple(pw.dims[n]) for n in pw.names],
                                             where='trained nets, it %d' % it, small64=g[p + 'grads_small_f64']))
        Y.check_values(learner.batch_data['batch_targets'].cpu().numpy()[::8], g[p + 'targets_sub'], g[p + 'targets_sub_f64'],
                       what='trained nets, targets')
        st = learner.get_stats()
        for k in ('value_mean', 'policy_total_loss', 'policy_gradient_norm', 'q_loss1', 'q_gradient_norm1', 'q_loss2', 'q_gradient_norm2'):
            np.testing.assert_allclose(st[k], g[p + k], rtol=5e-5, atol=1e-6, err_msg=k)
    assert int(pw.nonfinite.sum().item()) == 0 and pw.check_status() == 0
    print('trained nets (%s engine): worst error / allowance = %.2f' % (engine, worst))


@pytest.mark.parametrize('name,reps', [('c2_mpg_v2_B4096', 3000), ('c3_nadp_B8192', 600), ('c4_td3_B65536', 200)])
def test_repeated_launches_are_bit_identical(name, reps):
    """The same gradient computation launched `reps` times must return the same bits every time.  This is the check that
    exposed the lost packed-FMA products of the round-2 weight-gradient kernel (one 16-column run of
    dW1 off by a single row's product in most launches, invisible to a tolerance of 1e-4 and to a two-launch comparison)."""
    from mpg_amd.config import default_args
    from mpg_amd.learners import MPGLearner, NADPLearner, TD3Learner
    from mpg_amd.policy import PolicyWithQs
    from tests.golden_inputs import bench_case_inputs
    d = bench_case_inputs(name)
    B = d['B']
    flat = {k: np.concatenate([np.asarray(w).ravel() for w in v]).astype(np.float32) for k, v in d['nets'].items()}
    batch = [dev(x) for x in d['batch']]
    if d['kind'] == 'MPG-v2':
        learner, it, kw = MPGLearner(PolicyWithQs, default_args('MPG-v2', replay_batch_size=B, num_batch_reuse=1)), 100, dict(eps=dev(d['eps']))
    elif d['kind'] == 'NADP':
        learner, it, kw = NADPLearner(PolicyWithQs, default_args('NADP', replay_batch_size=B)), 0, dict(eps_q=dev(d['eps_q']), eps_pi=dev(d['eps_pi']))
    else:
        learner, it, kw = TD3Learner(PolicyWithQs, default_args('TD3', replay_batch_size=B)), 0, dict(smooth_eps=dev(d['smooth_eps']))
    pw = learner.policy_with_value
    w = np.concatenate([flat[n] for n in pw.names])
    pw.set_flat(w, (w * np.float32(0.97)).astype(np.float32))

    def run():
        learner.counter = 0
        return torch.cat([x.reshape(-1) for x in learner.compute_gradient(batch, None, None, it, **kw)])
    ref = torch.stack([run().clone() for _ in range(5)]).median(0).values
    differing = torch.stack([(run() != ref).any() for _ in range(reps)]).sum().item()
    assert differing == 0, '%d of %d launches differ from the majority result' % (differing, reps)


@pytest.mark.parametrize('case', ['MPG-v1 B=256', 'MPG-v2 K=3 B=256', 'MPG-v2 B=256'])
def test_repeated_launches_are_bit_identical_small_batches_and_look_ahead(case):
    """The same check on the paths the bench-size cases do not take: the reference's own batch size (C1:
    256 rows - the small-batch launch geometry: one-group target / critic launches, k_wgrad_multi with few groups per chunk),
    MPG-v1 (25 real-env steps + the n-step target in front of the gradient) and observations with look-ahead entries
    (num_future_data = 3: the launch-per-stage path with the 16-wide network kernels and the WIDE sweeps).  400 launches each."""
    from mpg_amd.config import default_args
    from mpg_amd.learners import MPGLearner
    from mpg_amd.policy import PolicyWithQs
    from tests.golden_inputs import reset_law_obs
    alg = 'MPG-v1' if 'v1' in case else 'MPG-v2'
    K = 3 if 'K=3' in case else 0
    B = 256
    rng = np.random.Generator(np.random.PCG64(len(case)))
    args = default_args(alg, replay_batch_size=B, num_batch_reuse=1, num_future_data=K)
    learner = MPGLearner(PolicyWithQs, args)
    obs = reset_law_obs(rng, B)
    if K:
        obs = np.concatenate([obs, rng.normal(0, 1, (B, K)).astype(np.float32)], 1)
    act = rng.uniform(-1, 1, (B, 2)).astype(np.float32)
    obs2 = obs + rng.normal(0, 0.05, obs.shape).astype(np.float32)
    batch = [dev(obs), dev(act), dev(rng.normal(-1, 1, B)), dev(obs2), dev(np.ones(B))]
    eps = dev(rng.standard_normal((25, B)))

    def run():
        learner.counter = 0
        return torch.cat([x.reshape(-1) for x in learner.compute_gradient(batch, None, None, 100, eps=eps)])
    ref = torch.stack([run().clone() for _ in range(5)]).median(0).values
    assert torch.isfinite(ref).all()
    differing = torch.stack([(run() != ref).any() for _ in range(400)]).sum().item()
    assert differing == 0, '%d of 400 launches differ from the majority result' % differing


def test_look_ahead_nine_and_ten_entries_are_served_and_eleven_refused():
    """path_tracking_env.py:385-402 accepts any num_future_data; here the env, the 16-wide policy kernels and the 24-wide critic kernels
    serve num_future_data <= 10 (include/mpg_hip.h, mpg_cfg_t.obs_dim <= 16) through worker, ring, learner and optimizer - round 4; the
    reference-generated golden at K = 10 is test_compute_gradient_vs_reference_golden[v2-10].  11 is refused with a defined error when
    the classes are constructed, not at the first launch."""
    from mpg_amd._lib import MpgError
    from mpg_amd.config import default_args
    from mpg_amd.envs import PathTrackingEnv
    from mpg_amd.policy import PolicyWithQs
    env = PathTrackingEnv(num_future_data=10, num_agent=8)
    assert env.reset().shape == (8, 16)
    for K in (9, 10):
        pw = PolicyWithQs(**vars(default_args('MPG-v2', num_future_data=K)))
        assert pw.obs_dim == 6 + K and pw.dims['Q1'][0] == 8 + K
    with pytest.raises(MpgError, match='num_future_data'):
        PolicyWithQs(**vars(default_args('MPG-v2', num_future_data=11)))
    with pytest.raises((MpgError, ValueError, AssertionError)):
        PathTrackingEnv(num_future_data=11, num_agent=8)


def test_replay_buffer_ring_and_gather_bit_exact():
    from mpg_amd.buffer import ReplayBuffer
    from mpg_amd.config import default_args
    args = default_args(max_buffer_size=1000, replay_starts=300, replay_batch_size=256)
    rb = ReplayBuffer(args, 0)
    rng = np.random.Generator(np.random.PCG64(0))
    mirror = {k: np.zeros(s, np.float32) for k, s in (('o', (1000, 6)), ('a', (1000, 2)), ('r', 1000), ('o2', (1000, 6)))}
    nxt, size = 0, 0
    assert rb.replay() is None                                     # buffer.py:85-86
    for n in (300, 512, 512):                                      # wraps around the 1000-slot ring
        o, a, r, o2 = [rng.standard_normal(s).astype(np.float32) for s in ((n, 6), (n, 2), n, (n, 6))]
        rb.add_batch((dev(o), dev(a), dev(r), dev(o2), torch.ones(n, dtype=torch.uint8, device=DEV)))
        pos = (nxt + np.arange(n)) % 1000
        mirror['o'][pos], mirror['a'][pos], mirror['r'][pos], mirror['o2'][pos] = o, a, r, o2
        nxt, size = (nxt + n) % 1000, min(size + n, 1000)
    assert len(rb) == 1000
    s = rb.replay()
    idx = s[-1].cpu().numpy()
    assert idx.min() >= 0 and idx.max() < 1000 and len(np.unique(idx)) > 200
    np.testing.assert_array_equal(s[0].cpu().numpy(), mirror['o'][idx])       # gather is a copy: bit-exact
    np.testing.assert_array_equal(s[1].cpu().numpy(), mirror['a'][idx])
    np.testing.assert_array_equal(s[2].cpu().numpy(), mirror['r'][idx])
    np.testing.assert_array_equal(s[3].cpu().numpy(), mirror['o2'][idx])
    assert bool((s[4] == 1).all())
    big = rb.sample_idxes(1 << 16).cpu().numpy()                               # uniformity of the index draw
    hist = np.bincount(big, minlength=1000)
    assert hist.min() > 20 and abs(hist.mean() - 65.536) < 1e-6 and hist.std() < 12


@pytest.mark.parametrize('alg', ['MPG-v2', 'MPG-v1'])
def test_single_process_training_loop_runs_and_learns_the_critic(alg):
    """SingleProcessOffPolicyOptimizer.step order (optimizer.py:330-362) for a few dozen iterations: finite, the
    policy optimizer ticks every 2nd iteration (delay_update), targets move by Polyak, critic loss goes down."""
    from mpg_amd.buffer import ReplayBuffer
    from mpg_amd.config import default_args
    from mpg_amd.learners import MPGLearner
    from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer
    from mpg_amd.policy import PolicyWithQs
    from mpg_amd.worker import OffPolicyWorker
    args = default_args(alg, num_agent=64, batch_size=512, replay_batch_size=256, replay_starts=1024,
                        max_buffer_size=8192, value_lr_schedule=[1e-3, 100000, 1e-4])
    worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
    learner = MPGLearner(PolicyWithQs, args)
    rb = ReplayBuffer(args, 0)
    opt = SingleProcessOffPolicyOptimizer(worker, learner, rb, None, args)
    assert len(rb) >= 1024
    pw = worker.policy_with_value
    t0 = pw.targets.clone()
    losses = []
    for i in range(40):
        opt.step()
        losses.append(learner.get_stats()['q_loss1'])
    assert all(np.isfinite(losses)) and torch.isfinite(pw.params).all()
    assert pw.opt_steps['Q1'] == 40 and pw.opt_steps['policy'] == 20
    assert (pw.targets - t0).abs().max().item() > 0
    assert np.mean(losses[-5:]) < np.mean(losses[:5])
    assert int(pw.nonfinite.sum().item()) == 0


def test_mpg_v1_with_look_ahead_observations_vs_oracle():
    """MPG-v1 at num_future_data = 2: the critic target comes from 25 REAL env steps (mpg_learner.py:109-124,146-169), whose
    observations carry the env's own look-ahead entries; gradient list against the oracle on the same minibatch and noise."""
    from mpg_amd.config import default_args
    from mpg_amd.learners import MPGLearner
    from mpg_amd.policy import PolicyWithQs
    from tests.golden_inputs import mlp_weights_flat, reset_law_obs
    K, B = 2, 64
    rng = np.random.Generator(np.random.PCG64(77))
    args = default_args('MPG-v1', replay_batch_size=B, num_batch_reuse=1, num_future_data=K)
    learner = MPGLearner(PolicyWithQs, args)
    pw = learner.policy_with_value
    w = {'Q1': mlp_weights_flat(rng, 8 + K, 1), 'policy': mlp_weights_flat(rng, 6 + K, 4)}
    flat = np.concatenate([w[n] for n in pw.names])
    pw.set_flat(flat, (flat * np.float32(0.97)).astype(np.float32))
    # a minibatch whose look-ahead entries are the env's: reset the device env from base observations and step it once
    from mpg_amd.envs import PathTrackingEnv
    env = PathTrackingEnv(num_future_data=K, num_agent=B)
    base = reset_law_obs(rng, B)
    env.reset(init_obs=dev(np.concatenate([base, np.zeros((B, K), np.float32)], 1)))
    obs = env.step(dev(rng.uniform(-0.3, 0.3, (B, 2)).astype(np.float32)))[0].clone()
    act = dev(rng.uniform(-1, 1, (B, 2)).astype(np.float32))
    obs2, rew, _, _ = env.step(act)
    batch = [obs, act, rew.clone(), obs2.clone(), torch.zeros(B, device=DEV)]
    eps = rng.standard_normal((25, B)).astype(np.float32)
    grads = learner.compute_gradient(batch, None, None, 100, eps=dev(eps))
    got = torch.cat([x.reshape(-1) for x in grads]).cpu().numpy()
    cfg = O.Cfg(obs_dim=6 + K, obs_scale=list(O.OBS_SCALE_PT) + [1.] * K)
    nb = [b.cpu().numpy() for b in batch]
    ref = {}
    for dt in (torch.float32, torch.float64):
        nets = O.Nets(cfg, w, target_scale=np.float32(0.97), dtype=dt)
        g, st = O.mpg_compute_gradient(cfg, nets, nb, eps, 100, 'MPG-v1')
        ref[dt] = np.concatenate([np.asarray(x, np.float64).ravel() for x in g])
    Y.check_gradients(got, ref[torch.float32].astype(np.float32), ref[torch.float64][::8], [(n,) + tuple(pw.dims[n]) for n in pw.names],
                      where='MPG-v1 K=2')


@pytest.mark.parametrize('fused,K', [(False, 3), (True, 3), (True, 10)])
def test_training_loop_with_look_ahead_observations(fused, K):
    """num_future_data = 3 (and 10: 16-wide policy, 18-wide critics - the 24-column network kernels) through worker, replay ring, learner and optimizer (train_script.py:90,146-147; worker.py:38;
    mpg_learner.py:35,48): observations are 9 wide, the first layers 9 / 11 wide.  The gradient of a minibatch of the worker's
    own transitions equals the oracle's on the same minibatch, weights and model noise; the loop runs, stays finite and learns
    the critic."""
    from mpg_amd.buffer import ReplayBuffer
    from mpg_amd.config import default_args
    from mpg_amd.learners import MPGLearner
    from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer
    from mpg_amd.policy import PolicyWithQs
    from mpg_amd.worker import OffPolicyWorker
    torch.manual_seed(0)                          # the model noise of the checked gradient (everything else is Philox-keyed)
    args = default_args('MPG-v2', num_agent=64, batch_size=512, replay_batch_size=256, replay_starts=1024, max_buffer_size=8192,
                        value_lr_schedule=[1e-3, 100000, 1e-4], num_future_data=K)
    assert args.obs_dim == 6 + K and len(args.obs_scale) == 6 + K
    worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
    learner = MPGLearner(PolicyWithQs, args)
    rb = ReplayBuffer(args, 0)
    opt = SingleProcessOffPolicyOptimizer(worker, learner, rb, None, args, fused=fused)
    assert (opt._fused is not None) == fused
    pw = worker.policy_with_value
    assert rb.obs.shape[1] == 6 + K and pw.dims['policy'] == (6 + K, 4) and pw.dims['Q1'] == (8 + K, 1)
    # the look-ahead entries in the ring are the env's (path_tracking_env.py:385-402), not copies of delta_y
    o = rb.obs[:len(rb)].cpu().numpy()
    assert np.abs(o[:, 6:] - o[:, 3:4]).max() > 1e-3
    # the gradient of a minibatch of the worker's own transitions (look-ahead entries computed by the env), with recorded model
    # noise, against the oracle - on the freshly initialised networks, like every fixture: a few dozen iterations into training
    # the look-ahead columns make the policy gradient a sum of nearly cancelling terms and a single float32 run is a noisy
    # yard-stick (24 seeds scanned: the float32 oracle itself 2e-6 ... 8e-5 from the float64 one, this engine 0.1 ... 5 x that;
    # tools/diag/lookahead_error_scan.py)
    batch = [b.clone() for b in rb.sample(256)[:5]]
    eps = torch.randn(25, 256, device=DEV)
    grads = learner.compute_gradient(batch, None, None, 500, eps=eps)
    got = torch.cat([x.reshape(-1) for x in grads]).cpu().numpy()
    learner.counter = 0
    cfg = O.Cfg(obs_dim=6 + K, obs_scale=list(O.OBS_SCALE_PT) + [1.] * K)
    flat, tflat = pw.params.cpu().numpy(), pw.targets.cpu().numpy()
    off = np.cumsum([0] + list(pw.sizes))
    w = {n: flat[off[i]:off[i + 1]] for i, n in enumerate(pw.names)}
    wt = {n: tflat[off[i]:off[i + 1]] for i, n in enumerate(pw.names)}
    nb = [b.cpu().numpy() for b in batch]
    ref = {}
    for dt in (torch.float32, torch.float64):
        nets = O.Nets(cfg, w, flat_targets=wt, dtype=dt)
        g, _ = O.mpg_compute_gradient(cfg, nets, nb, eps.cpu().numpy(), 500, 'MPG-v2')
        ref[dt] = np.concatenate([np.asarray(x, np.float64).ravel() for x in g])
    Y.check_gradients(got, ref[torch.float32].astype(np.float32), ref[torch.float64][::8], [(n,) + tuple(pw.dims[n]) for n in pw.names],
                      where='look-ahead K=3 (%s)' % ('native step driver' if fused else 'method by method'))
    # then the loop itself: runs, stays finite, the critic learns
    losses = []
    for i in range(30):
        opt.step()
        losses.append(learner.get_stats()['q_loss1'])
    assert all(np.isfinite(losses)) and torch.isfinite(pw.params).all()
    assert pw.opt_steps['Q1'] == 30 and pw.opt_steps['policy'] == 15
    assert np.mean(losses[-5:]) < np.mean(losses[:5])
    pw.check_status()


@pytest.mark.parametrize('alg', ['MPG-v2', 'MPG-v1'])
def test_native_step_driver_equals_method_by_method_path(alg):
    """mpg_step_begin/_end enqueue the same launches as the python classes: after the same number of iterations the
    counters and the replay ring are identical, and parameters / Adam moments / targets agree to float32 rounding (the
    host-side scalars - rule-based weights, bias-corrected learning rates - are evaluated by libm in one path and by
    numpy in the other, which may differ in the last bit)."""
    from mpg_amd.buffer import ReplayBuffer
    from mpg_amd.config import default_args
    from mpg_amd.learners import MPGLearner
    from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer
    from mpg_amd.policy import PolicyWithQs
    from mpg_amd.worker import OffPolicyWorker

    def run(fused):
        args = default_args(alg, num_agent=64, batch_size=128, replay_batch_size=96, replay_starts=512, max_buffer_size=1000,
                            num_batch_reuse=2 if alg == 'MPG-v1' else 1)
        worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
        learner = MPGLearner(PolicyWithQs, args)
        rb = ReplayBuffer(args, 0)
        opt = SingleProcessOffPolicyOptimizer(worker, learner, rb, None, args, sampling_interval=3, fused=fused)
        assert (opt._fused is not None) == fused
        for _ in range(9):
            opt.step()
        pw = worker.policy_with_value
        st = learner.get_stats()
        return [pw.params.clone(), pw.targets.clone(), pw.m.clone(), pw.v.clone(), rb.obs.clone(), rb.rew.clone(),
                learner.flat.clone()], (dict(pw.opt_steps), rb._next_idx, len(rb), worker._noise_ctr, st['q_loss1'])
    a, ca = run(True)
    b, cb = run(False)
    assert ca[:4] == cb[:4], (ca, cb)
    assert abs(ca[4] - cb[4]) <= 1e-6 * abs(cb[4])
    assert torch.equal(a[4], b[4])                    # ring: same reset-law draws in the same slots
    assert (a[5] - b[5]).abs().max().item() < 1e-4    # rewards depend on the (rounding-different) policy
    for x, y in zip(a[:4], b[:4]):
        assert (x - y).abs().max().item() <= 1e-6 * max(1.0, y.abs().max().item())
    g1, g2 = a[6], b[6]
    assert ((g1 - g2).norm() / g2.norm()).item() < 1e-5


def test_stock_methods_interleaved_with_native_steps():
    """The stock worker / buffer methods may be called between native steps (mpg_amd/fused.py: sync_in / push): a run that
    interleaves `worker.sample()` + `rb.add_batch()` with fused steps ends in the same ring, counters and parameters as
    the pure method-by-method path doing the same calls - no ring slot is overwritten, no Philox (seed, counter) pair is
    used twice."""
    from mpg_amd.buffer import ReplayBuffer
    from mpg_amd.config import default_args
    from mpg_amd.learners import MPGLearner
    from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer
    from mpg_amd.policy import PolicyWithQs
    from mpg_amd.worker import OffPolicyWorker

    def run(fused):
        args = default_args('MPG-v2', num_agent=64, batch_size=64, replay_batch_size=96, replay_starts=256, max_buffer_size=700)
        worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
        learner = MPGLearner(PolicyWithQs, args)
        rb = ReplayBuffer(args, 0)
        opt = SingleProcessOffPolicyOptimizer(worker, learner, rb, None, args, sampling_interval=2, fused=fused)
        assert (opt._fused is not None) == fused
        for it in range(12):
            opt.step()
            if it % 3 == 1:                      # extra samples through the stock methods (wraps the 700-slot ring)
                batch, n = worker.sample_with_count()
                rb.add_batch(batch)
        pw = worker.policy_with_value
        return [pw.params.clone(), pw.targets.clone(), rb.obs.clone(), rb.act.clone(), worker.obs.clone()], \
            (dict(pw.opt_steps), rb._next_idx, len(rb), rb.replay_times, worker._noise_ctr, worker.env._ctr)
    a, ca = run(True)
    b, cb = run(False)
    assert ca == cb, (ca, cb)
    assert torch.equal(a[2], b[2]) and torch.equal(a[4], b[4])          # ring observations and the worker's current obs
    assert (a[3] - b[3]).abs().max().item() < 1e-4                      # stored actions depend on the policy (rounding)
    for x, y in zip(a[:2], b[:2]):
        assert (x - y).abs().max().item() <= 1e-6 * max(1.0, y.abs().max().item())


def test_replay_sample_uniform_equals_indices_plus_gather():
    import mpg_amd._lib as L
    from mpg_amd.buffer import ReplayBuffer
    from mpg_amd.config import default_args
    args = default_args(max_buffer_size=5000, replay_starts=100, replay_batch_size=777)
    rb = ReplayBuffer(args, 3)
    g = torch.Generator().manual_seed(0)
    n = 3000
    rb.add_batch((torch.randn(n, 6, generator=g).cuda(), torch.randn(n, 2, generator=g).cuda(), torch.randn(n, generator=g).cuda(),
                  torch.randn(n, 6, generator=g).cuda(), torch.ones(n, dtype=torch.uint8).cuda()))
    ref = rb.replay()                                   # mpg_uniform_indices + mpg_replay_gather, ctr = replay_times = 1
    B = 777
    out = [torch.empty(B, 6).cuda(), torch.empty(B, 2).cuda(), torch.empty(B).cuda(), torch.empty(B, 6).cuda(), torch.empty(B).cuda()]
    idx = torch.empty(B, dtype=torch.int32).cuda()
    L.call('mpg_replay_sample_uniform', L.c_int(len(rb)), L.c_int(B), L.c_u64(rb.seed), L.c_u64(1), L.c_int(6), L.c_int(2),
           L.ptr(rb.obs), L.ptr(rb.act), L.ptr(rb.rew), L.ptr(rb.obs2), L.ptr(rb.done), L.ptr(idx), *[L.ptr(t) for t in out],
           L.stream())
    assert torch.equal(idx, ref[-1])
    for x, y in zip(out, ref[:5]):
        assert torch.equal(x, y)


def test_fast_path_still_learns_path_tracking():
    """Learning-curve check: MPG-v2 with the reference's default learning rates, 2000 iterations of
    the native step driver (< 2 s), evaluated like evaluator.py:118-184 (200-step deterministic episodes).  A randomly
    initialised policy scores about -4000 per episode; the reference's plots draw their base line at -30 (ploter.py:85)."""
    from mpg_amd.buffer import ReplayBuffer
    from mpg_amd.config import default_args
    from mpg_amd.evaluator import Evaluator
    from mpg_amd.learners import MPGLearner
    from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer
    from mpg_amd.policy import PolicyWithQs
    from mpg_amd.worker import OffPolicyWorker
    args = default_args('MPG-v2', num_agent=256, batch_size=256, replay_batch_size=256, replay_starts=3000, num_eval_agent=256,
                        rule_based_bias_total_ite=6000)
    worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
    learner = MPGLearner(PolicyWithQs, args)
    rb = ReplayBuffer(args, 0)
    opt = SingleProcessOffPolicyOptimizer(worker, learner, rb, None, args, sampling_interval=1)
    ev = Evaluator(PolicyWithQs, args.env_id, args)
    ev.share_policy(worker.policy_with_value)
    before = ev.run_evaluation(0)
    for _ in range(2000):
        opt.step()
    after = ev.run_evaluation(2000)
    assert before['episode_return'] < -1000
    assert after['episode_return'] > -300 and after['delta_y_mse'] < 5.0 and after['delta_phi_mse'] < 0.2, (before, after)
    assert int(worker.policy_with_value.nonfinite.sum().item()) == 0


@pytest.mark.parametrize('K', [0, 3])
def test_deriv_interval_policy_learner_path(golden, K):
    """deriv_interval_policy=True (mpg_learner.py:247-248): the learner routes through the fine-grained entry points;
    critic gradients are the default path's, the policy gradient is the clipped full-BPTT gradient of the same loss.
    K = 3: the same with look-ahead observations (round 4: any combination the reference accepts)."""
    g = golden('mpg_v2_H256_B64%s.npz' % ('_K%d' % K if K else ''))
    base = _learner(g, 'v2', K=K)
    batch = [dev(g[k]) for k in ('batch_obs', 'batch_actions', 'batch_rewards', 'batch_obs_tp1', 'batch_dones')]
    ref = torch.cat([x.reshape(-1) for x in base.compute_gradient(batch, None, None, 100, eps=dev(g['eps']))]).cpu().numpy()
    full = _learner(g, 'v2', K=K)
    full.deriv_interval_policy = True
    got = torch.cat([x.reshape(-1) for x in full.compute_gradient(batch, None, None, 100, eps=dev(g['eps']))]).cpu().numpy()
    pw = full.policy_with_value
    nq = pw.offsets[2]
    assert rel_l2(got[:nq], ref[:nq]) < 2e-6
    ocfg = O.Cfg(obs_dim=6 + K, obs_scale=list(O.OBS_SCALE_PT) + [1.] * K)
    nets = O.Nets(ocfg, {n: g['w_' + n] for n in pw.names}, dtype=torch.float64)
    reduced, _, _ = O.model_rollout_for_policy_update(ocfg, nets, torch.as_tensor(g['batch_obs']).double(),
                                                      torch.as_tensor(g['eps']).double(), rollout_policy='policy')
    ws = O.rule_based_weights(100, ocfg.total_ite, ocfg.eta, ocfg.select, torch.float64)
    loss = torch.sum(ws * torch.stack([-reduced[k] for k in ocfg.select]))
    pg, _ = O.clip_by_global_norm(list(torch.autograd.grad(loss, nets.w['policy'])), ocfg.clip)
    want = np.concatenate([x.numpy().ravel() for x in pg])
    assert rel_l2(got[nq:], want) < 1e-4
    np.testing.assert_allclose(full.get_stats()['all_losses'], base.get_stats()['all_losses'], rtol=1e-5)


@pytest.mark.parametrize('alg,fused,per', [('MPG-v2', True, False), ('MPG-v1', True, False), ('MPG-v2', False, False),
                                           ('TD3', False, True)])
def test_checkpoint_resume_is_bit_identical(tmp_path, alg, fused, per):
    """save after 12 iterations, continue 9 more; a freshly built optimizer that loads the file and
    runs the same 9 iterations ends with bit-identical parameters, targets, Adam moments and replay ring."""
    from mpg_amd.buffer import PrioritizedReplayBuffer, ReplayBuffer
    from mpg_amd.checkpoint import load_checkpoint, save_checkpoint
    from mpg_amd.config import default_args
    from mpg_amd.learners import MPGLearner, TD3Learner
    from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer
    from mpg_amd.policy import PolicyWithQs
    from mpg_amd.worker import OffPolicyWorker

    def build(seed):
        args = default_args(alg, num_agent=64, batch_size=64, replay_batch_size=128, replay_starts=256, max_buffer_size=1024,
                            seed=seed, buffer_type='priority' if per else 'normal')
        worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
        learner = (TD3Learner if alg == 'TD3' else MPGLearner)(PolicyWithQs, args)
        rb = (PrioritizedReplayBuffer if per else ReplayBuffer)(args, 0)
        return SingleProcessOffPolicyOptimizer(worker, learner, rb, None, args, sampling_interval=3, fused=fused)

    def snapshot(opt):
        pw = opt.worker.policy_with_value
        return [t.clone() for t in (pw.params, pw.targets, pw.m, pw.v, opt.replay_buffer.obs, opt.replay_buffer.rew,
                                    opt.worker.obs)]

    a = build(seed=5)
    assert (a._fused is not None) == fused
    for _ in range(12):
        a.step()
    path = save_checkpoint(str(tmp_path / 'ckpt.npz'), a)
    for _ in range(9):
        a.step()
    b = build(seed=99)                   # different seed: every stream must come from the file
    meta = load_checkpoint(path, b)
    assert meta['optimizer']['iteration'] == 12 and b.iteration == 12
    for _ in range(9):
        b.step()
    for x, y in zip(snapshot(a), snapshot(b)):
        assert torch.equal(x, y)
    assert a.worker.policy_with_value.opt_steps == b.worker.policy_with_value.opt_steps
    assert a.num_sampled_steps == b.num_sampled_steps
    # the file is a plain .npz: readable without the package
    z = np.load(path)
    assert z['policy/params'].dtype == np.float32 and z['buffer/obs'].shape[1] == 6
