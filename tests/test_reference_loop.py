"""The oracle's loop restatements (oracle.apply_gradients / AdamState, tests/c2_loop.py, tests/c3_loop.py) against fixtures that the
reference's OWN, unmodified loop code produced (tests/golden/make_golden.py, round 6):

    apply_gradients_ref.npz   PolicyWithQs.apply_gradients             policy.py:123-171, schedules :54-70
    worker_sample_ref.npz     OffPolicyWorker.sample                   worker.py:91-119
    loop_v2_ref.npz           SingleProcessOffPolicyOptimizer.step x 20, MPG-v2 at the reference's defaults   optimizer.py:286-397
    loop_nadp_ref.npz         the same with NADPLearner on the pendulum model                                 learners/nadp.py:209-241

Every random draw of those runs was the oracle's restatement of the device's Philox stream for the same event (DeviceStreams in the
generator), so the restated loops - and, in tests/test_reference_loop_gpu.py, the device loops - can be compared with them iteration
for iteration.  What is pinned: which optimizer steps when, the per-optimizer counters, which targets move when, the exploration noise
behind tanh, the reset after every step, the sample-every-10th order, the replay cadence.  What is NOT: Keras Adam's
arithmetic - the stand-in Adam is the published ApplyAdam formula in float32 torch ops, the oracle's is the same formula in numpy
with the step size formed in float64; they agree to float32 rounding, which is the bar below.

No GPU needed."""
import numpy as np
import pytest
import torch

from oracle import mpg_oracle as O
from tests.golden_inputs import LOOP_SEED, NET_DIMS, apply_case_grads, loop_case_weights


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def case_cfg(case):
    return O.Cfg(env='InvertedPendulumConti-v0', select=[25], delay_update=1) if case == 'nadp' else O.Cfg()


@pytest.mark.parametrize('case', ['v2', 'v1', 'nadp'])
def test_apply_gradients_vs_reference_policy_with_qs(golden, case):
    """oracle.apply_gradients x 6 on the fixture's gradient lists against the reference's PolicyWithQs.apply_gradients.
    Exact: the per-optimizer `iterations` after every call (critics every call; policy on iterations 0, 2, 4 with delay_update 2,
    every call with 1), WHICH entries moved (a target is untouched on odd iterations with delay 2).
    Bars: parameter and target UPDATES within 2e-6 relative L2 of the reference's (float32 rounding of the step size:
    lr_t is formed in float64 here, in float32 by the stand-in), and within 4 x the reference's own float32-vs-float64 gap + 1e-6."""
    g = golden('apply_gradients_ref.npz')
    names = [n for n, _, _ in NET_DIMS[case]]
    cfg = case_cfg(case)
    w = loop_case_weights(case)
    w0 = np.concatenate([w[n] for n in names])
    tgt = {k: v.copy() for k, v in w.items()}
    opt = {k: O.AdamState(v.size) for k, v in w.items()}
    grads = apply_case_grads(case, int(g['n_iter']))
    prev_t = w0.copy()
    for it in range(int(g['n_iter'])):
        O.apply_gradients(cfg, w, tgt, opt, grads[it], it, names)
        assert [opt[n].step for n in names] == list(g[case + '_opt_iterations'][it]), it
        p, t = np.concatenate([w[n] for n in names]), np.concatenate([tgt[n] for n in names])
        ref_p, ref_t = g[case + '_params_sub'][it], g[case + '_targets_sub'][it]
        moved = not np.array_equal(t, prev_t)
        assert moved == (it % cfg.delay_update == 0), (it, 'targets move only on delayed iterations (policy.py:136-153)')
        assert moved == (not np.array_equal(ref_t, g[case + '_targets_sub'][it - 1] if it else w0[::64]))
        prev_t = t.copy()
        assert rel_l2(p[::64] - w0[::64], ref_p - w0[::64]) <= 2e-6, it
        assert rel_l2(t[::64] - w0[::64], ref_t - w0[::64]) <= 2e-6 or not np.any(ref_t - w0[::64]), it
    every = int(g[case + '_every'])
    e_p = rel_l2((p - w0)[::every], g[case + '_params'] - w0[::every])
    e_t = rel_l2((t - w0)[::every], g[case + '_targets'] - w0[::every])
    # yard-stick: the reference graph in float64
    u64, tu64 = g[case + '_update_f64'], g[case + '_target_update_f64']
    k4 = 4 // every                                          # the float32 finals are stored every `every`-th entry, every in (1, 4)
    gap = rel_l2(g[case + '_params'][::k4] - w0[::4], u64)
    gap_t = rel_l2(g[case + '_targets'][::k4] - w0[::4], tu64)
    mine, mine_t = rel_l2((p - w0)[::4], u64), rel_l2((t - w0)[::4], tu64)
    print('%s: update vs reference float32 %.2e (targets %.2e); vs float64: %.2e / %.2e, reference float32 %.2e / %.2e' %
          (case, e_p, e_t, mine, mine_t, gap, gap_t))
    assert e_p <= 2e-6 and e_t <= 2e-6
    assert mine <= 4 * gap + 1e-6 and mine_t <= 4 * gap_t + 1e-6


def test_worker_sample_vs_reference_worker(golden):
    """tests/c2_loop.py's sample() (the oracle's env, policy and the Philox restatements) against the reference's
    OffPolicyWorker.sample run on the same streams: 2 calls x 8 steps x 8 agents.  Actions (policy + exploration noise behind tanh,
    un-clipped: the env clips its own copy, worker.py:98,108), raw rewards, next observations, the always-true done flag, the
    reset draw after every step (counters), the observation the worker is left holding."""
    from tests.c2_loop import OracleConfig2Loop
    g = golden('worker_sample_ref.npz')
    NA, BS, calls = int(g['num_agent']), int(g['batch_size']), int(g['calls'])
    loop = OracleConfig2Loop(loop_case_weights('v2'), seed=int(g['seed']), num_agent=NA, batch_size=BS, replay_starts=calls * BS, capacity=4096)
    n = calls * BS
    assert loop.size == n == g['obs'].shape[0]
    assert [loop.env_ctr, loop.noise_ctr] == list(g['counters'])
    r = loop.ring
    for k, tol in (('obs', 2e-5), ('act', 2e-6), ('obs2', 2e-5), ('rew', 2e-5)):
        ref, ref64 = g[k], g[k + '_f64']
        err = np.abs(r[k][:n] - ref).max()
        gap = np.abs(ref - ref64).max()
        print('%-4s max abs err vs reference float32 %.2e (reference float32 vs float64 %.2e)' % (k, err, gap))
        assert err <= tol * max(1.0, np.abs(ref).max()), k
    assert np.array_equal(r['done'][:n] != 0, g['done'] != 0) and g['done'].all()
    np.testing.assert_allclose(loop.env.obs, g['final_obs'], rtol=2e-5, atol=2e-5)


def _check_loop_against_fixture(g, loop, names, idx_of, stat_of, n_iter, w0, tol_update=1e-3):
    """shared by the two loop tests: step the restated loop n_iter times beside the fixture of the reference's own loop"""
    assert loop.size == int(g['fill']), 'ring after the constructor fill (optimizer.py:310-313)'
    worst = 0.0
    for it in range(n_iter):
        loop.step()
        if idx_of(loop) is not None:
            np.testing.assert_array_equal(idx_of(loop), g['idx'][it], err_msg='replay indices, iteration %d' % it)
        assert [loop.opt[n].step for n in names] == list(g['opt_iterations'][it]), it
        p = np.concatenate([loop.w[n] for n in names])
        for k, key in enumerate(g['stat_keys']):
            got = stat_of(loop, str(key))
            if got is not None:
                ref = g['stats'][it][k]
                assert abs(got - ref) <= 2e-4 * abs(ref) + 1e-6, (it, key, got, ref)
        o = 0
        for k, n in enumerate(names):
            sz = loop.w[n].size
            un = np.linalg.norm(p[o:o + sz].astype(np.float64) - w0[o:o + sz])
            assert abs(un - g['update_norms'][it][k]) <= 1e-3 * g['update_norms'][it][k] + 1e-12, (it, n, un, g['update_norms'][it][k])
            o += sz
        e = rel_l2(p[::64] - w0[::64], g['params_sub'][it] - w0[::64])
        worst = max(worst, e)
        assert e <= tol_update, (it, e)
    return worst


@pytest.mark.parametrize('case', ['v2', 'td3', 'v1', 'v2k3'])
def test_config2_loop_restatement_vs_reference_optimizer(golden, case):
    """[td3: the same with the reference's TD3Learner (learners/td3.py:150-188, uniform replay; the smoothing noise td3.py:74 on the
    restated mpg_normal_fill stream).  v1: MPGLearner MPG-v1 - networks [Q1 | policy], the critic's target = the 25-step REAL-env return of
    the learner's own 256-agent env (mpg_learner.py:109-124,146-169), recomputed with a new minibatch every num_batch_reuse = 10 calls.
    v2k3: MPG-v2 with num_future_data = 3 - nine-entry observations (three look-ahead delta_y terms, path_tracking_env.py:385-402) through
    env, worker, ring, model (:262-268) and learner.]
    tests/c2_loop.py x 20 iterations against the reference's SingleProcessOffPolicyOptimizer + OffPolicyWorker + ReplayBuffer +
    MPGLearner (MPG-v2) + PolicyWithQs at the reference's defaults (8 agents, 512 transitions per sample, replay_starts 3000, batch
    256, sampling at iterations 0 and 10, delay_update 2), same Philox inputs, same initial weights.
    Exact: ring length after the fill and at the end, the replay indices of every iteration (they depend on the ring length, i.e. on
    the sampling cadence), the three optimizers' counters after every iteration, the env / noise / replay / learner counters at the end.
    Bars: learner statistics 2e-4 relative; per-network update norms 1e-3; parameter update (every 64th entry, every iteration, and
    all entries at the end) 1e-3 relative L2 and within 4 x the reference's own float32-vs-float64 gap; ring contents 2e-5."""
    from tests.c2_loop import OracleConfig2Loop
    g = golden('loop_%s_ref.npz' % case)
    dims = case if case in ('v1', 'v2k3') else 'v2'
    names = [n for n, _, _ in NET_DIMS[dims]]
    w = loop_case_weights(dims)
    w0 = np.concatenate([w[n] for n in names])
    nthreads = torch.get_num_threads()
    torch.set_num_threads(4)
    loop = OracleConfig2Loop(w, seed=int(g['seed']), num_agent=8, batch_size=512, replay_batch_size=256, replay_starts=3000,
                             capacity=8192, sampling_interval=10, alg={'v2': 'MPG-v2', 'td3': 'TD3', 'v1': 'MPG-v1', 'v2k3': 'MPG-v2'}[case],
                             num_future_data=3 if case == 'v2k3' else 0)
    n_iter = int(g['n_iter'])
    stat = lambda lp, key: float(np.asarray(lp.stats[key])) if key in lp.stats else None
    # (MPG-v1: the reference's buffer draws every iteration, the learner takes a new minibatch every 10th call - the restated loop draws
    # only those; the Philox counter is the replay count in both, so the draws it does make are the fixture's rows 0 and 10)
    idx_of = lambda lp: lp.idx if (lp.counter - 1) % lp.reuse == 0 else None
    worst = _check_loop_against_fixture(g, loop, names, idx_of, stat, n_iter, w0)
    torch.set_num_threads(nthreads)
    n = int(g['ring_len'])
    assert loop.size == n and loop.next == int(g['ring_next']) and loop.replay_times == int(g['replay_times']) == n_iter
    assert loop.counter == int(g['learner_counter'])
    assert [loop.env_ctr, loop.noise_ctr] == list(g['counters'][:2])
    for k in ('obs', 'act', 'rew', 'obs2'):
        ref = g['ring_' + k]
        assert np.abs(loop.ring[k][:n] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), k
    assert loop.ring['done'][:n].all() and g['ring_done'].all()
    p, t = loop.flat()
    e_p, e_t = rel_l2(p - w0, g['params'] - w0), rel_l2(t - w0, g['targets'] - w0)
    gap, mine = rel_l2((g['params'] - w0)[::4], g['update_f64']), rel_l2((p - w0)[::4], g['update_f64'])
    print('config-2 loop restatement (%s) vs the reference optimizer, %d iterations: update %.2e (targets %.2e; worst sampled %.2e); '
          'vs float64 %.2e, reference float32 %.2e' % (case, n_iter, e_p, e_t, worst, mine, gap))
    assert e_p <= 1e-3 and e_t <= 1e-3 and mine <= 4 * gap + 1e-6


def test_config3_loop_restatement_vs_reference_optimizer(golden):
    """tests/c3_loop.py x 20 iterations against the reference's SingleProcessOffPolicyOptimizer + OffPolicyWorker (one agent behind
    the reference's DummyVecEnv) + ReplayBuffer + NADPLearner on the pendulum MODEL + PolicyWithQs (delay_update 1) at
    train_script4mujoco.py's defaults.  The real environment under the worker is the oracle's closed-form cart-pole in both runs (the
    reference's is MuJoCo: absent, parity of that env unpinned) - everything around it is the reference's code.  Same bars as the
    config-2 test; the reset semantics pinned here are DummyVecEnv's (utils/dummy_vec_env.py:33-37: reset only when done)."""
    from tests.c3_loop import OracleConfig3Loop
    g = golden('loop_nadp_ref.npz')
    names = [n for n, _, _ in NET_DIMS['nadp']]
    w = loop_case_weights('nadp')
    w0 = np.concatenate([w[n] for n in names])
    nthreads = torch.get_num_threads()
    torch.set_num_threads(4)
    loop = OracleConfig3Loop(w['Q1'], w['policy'], seed=int(g['seed']), num_agent=1, batch_size=512, replay_batch_size=256,
                             replay_starts=3000, capacity=8192, sampling_interval=10)
    n_iter = int(g['n_iter'])
    stat = lambda lp, key: float(np.asarray(lp.stats[key])) if key in lp.stats else None
    worst = _check_loop_against_fixture(g, loop, names, lambda lp: lp.idx, stat, n_iter, w0)
    torch.set_num_threads(nthreads)
    n = int(g['ring_len'])
    assert loop.size == n and loop.next == int(g['ring_next']) and loop.replay_times == int(g['replay_times']) == n_iter
    assert loop.counter == int(g['learner_counter'])
    np.testing.assert_allclose(loop.ring_obs[:n], g['ring_obs'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(loop.ring_act[:n], g['ring_act'], rtol=1e-4, atol=2e-5)
    p, t = loop.flat()
    e_p, e_t = rel_l2(p - w0, g['params'] - w0), rel_l2(t - w0, g['targets'] - w0)
    gap, mine = rel_l2((g['params'] - w0)[::4], g['update_f64']), rel_l2((p - w0)[::4], g['update_f64'])
    print('config-3 loop restatement vs the reference optimizer, %d iterations: update %.2e (targets %.2e; worst sampled %.2e); '
          'vs float64 %.2e, reference float32 %.2e' % (n_iter, e_p, e_t, worst, mine, gap))
    assert e_p <= 1e-3 and e_t <= 1e-3 and mine <= 4 * gap + 1e-6


def test_config4_oracle_loop_pieces_vs_the_pinned_segment_tree(golden):
    """The vectorised tree pieces the teacher-forced config-4 loop (tests/c2_loop.py:OracleConfig4Loop) is made of, against
    O.SegmentTreeOracle - itself pinned bit for bit to the reference's utils/segment_tree.py by segment_tree_ref.npz:
    heap_tree == the tree after sequential __setitem__ calls (every node), find_prefixsum_idx_batch == find_prefixsum_idx, the IS
    weights == O.per_is_weights; and the loop runs free (forced indices = its own) for three iterations."""
    from tests.c2_loop import OracleConfig4Loop
    g = golden('segment_tree_ref.npz')
    cap, n, alpha = int(g['capacity']), int(g['n']), float(g['alpha'])
    st, mt = O.SegmentTreeOracle(cap, lambda a, b: a + b, 0.0), O.SegmentTreeOracle(cap, min, float('inf'))
    leaves = np.zeros(cap)
    for i, p in enumerate(g['prios']):
        st.set(i, float(p) ** alpha)
        mt.set(i, float(p) ** alpha)
        leaves[i] = float(p) ** alpha
    t = O.heap_tree(leaves, np.add)
    np.testing.assert_array_equal(t[1:], st.v[1:])
    tm = O.heap_tree(np.where(np.arange(cap) < n, leaves, np.inf), np.minimum)
    np.testing.assert_array_equal(tm[1:], mt.v[1:])
    np.testing.assert_array_equal(O.find_prefixsum_idx_batch(t, g['u'] * t[1]), g['idx'])
    u = O.per_uniform_philox(4096, 21, 5)
    assert 0.0 < u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.02
    torch.set_num_threads(4)
    loop = OracleConfig4Loop(loop_case_weights('v2'), seed=LOOP_SEED, num_agent=16, batch_size=16, replay_batch_size=64, replay_starts=128,
                             capacity=1000)
    assert loop.tree_cap == 1024 and loop.leaves[:128].min() == 1.0 and loop.leaves[128:].max() == 0.0
    for _ in range(3):
        st, _ = loop.trees()      # (any valid indices will do as the forced draw: here what the tree BEFORE this step's 16 additions gives)
        own = O.find_prefixsum_idx_batch(st, O.per_uniform_philox(64, loop.rb_seed, loop.replay_times + 1) * st[1])
        loop.step(np.minimum(own, loop.size - 1))
        w = loop.weights
        assert w.max() <= 1.0 + 1e-12 and w.min() > 0
    assert loop.max_priority >= 1.0 and loop.size == 128 + 3 * 16
    assert np.count_nonzero(loop.leaves) == loop.size


def test_per_oracle_pieces_vs_reference_prioritized_buffer(golden):
    """The oracle's PER pieces (heap_tree, find_prefixsum_idx_batch, the IS-weight formula of OracleConfig4Loop, per_is_weights over
    SegmentTreeOracle) against the reference's OWN PrioritizedReplayBuffer methods run unmodified (per_buffer_ref.npz: add at max
    priority, _sample_proportional, sample_with_weights_and_idxes, update_priorities with duplicates, a wrapping ring) - the shipped
    constructor and add_batch's weight 0 are dead code and are bypassed, see the generator.  Everything float64: leaves, every drawn
    index, totals and minima bit for bit; weights to 1e-15."""
    g = golden('per_buffer_ref.npz')
    cap, tcap, B, alpha, beta, eps = int(g['capacity']), int(g['tree_capacity']), int(g['B']), float(g['alpha']), float(g['beta']), float(g['eps'])
    leaves = np.zeros(tcap)
    seen = np.zeros(tcap, bool)
    max_p = 1.0

    def add(lo, hi, nxt):
        for i in range(lo, hi):
            leaves[nxt] = max_p ** alpha
            seen[nxt] = True
            nxt = (nxt + 1) % cap
        return nxt

    def draw(k, n_storage):
        st = O.heap_tree(leaves, np.add)
        mt = O.heap_tree(np.where(seen, leaves, np.inf), np.minimum)
        assert st[1] == float(g['draw%d_total' % k]) and mt[1] == float(g['draw%d_min' % k])
        idx = O.find_prefixsum_idx_batch(st, g['u'][k] * st[1])
        np.testing.assert_array_equal(idx, g['draw%d_idx' % k])
        w = (st[tcap + idx] / st[1] * n_storage) ** (-beta) / ((mt[1] / st[1] * n_storage) ** (-beta))
        np.testing.assert_allclose(w, g['draw%d_weights' % k], rtol=1e-15)
        return idx

    def update(idx, td):
        nonlocal max_p
        for j, x in zip(idx, td):                       # sequential: the last duplicate wins
            p = abs(float(x)) + eps
            leaves[j] = p ** alpha
            max_p = max(max_p, p)
    nxt = add(0, 500, 0)
    np.testing.assert_array_equal(leaves, g['leaves_a'])
    idx = draw(0, 500)
    update(idx, g['td'][0])
    np.testing.assert_array_equal(leaves, g['leaves_b'])
    assert max_p == float(g['max_priority_b'])
    nxt = add(500, 800, nxt)
    assert nxt == int(g['next_idx_c']) and int(g['len_c']) == cap
    np.testing.assert_array_equal(leaves, g['leaves_c'])
    idx = draw(1, cap)
    update(g['update2_idx'], g['td'][1])
    np.testing.assert_array_equal(leaves, g['leaves_d'])
    assert max_p == float(g['max_priority_d'])
    draw(2, cap)
