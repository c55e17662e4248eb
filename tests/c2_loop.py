"""The BENCH workload's training loop (train_scripts/train_script.py: OffPolicyWorker on PathTrackingEnv -> ReplayBuffer -> MPGLearner
(MPG-v2) -> clip / Adam / Polyak in SingleProcessOffPolicyOptimizer.step's order, optimizer.py:330-362) as a pure ORACLE run that
consumes EXACTLY the random inputs the device loop consumes - the twin of tests/c3_loop.py for configs 1 / 2.  The device's random
inputs are counter-based Philox draws (oracle/mpg_oracle.py restates them): the reset law of every agent after every step (`done` is
always true, SURVEY B-0), the exploration noise behind the policy (worker.py:97-98), the uniform replay indices, the model noise of
the 25-step rollout.

Seeds and counters as mpg_amd derives them from `args.seed`:
    worker / env seed = seed * 1000003 + worker_id     exploration noise counter: one per policy call; reset counter: one per env.reset()
    replay seed       = seed * 7919 + buffer_id        counter = replay_times (incremented before the draw)
    learner seed      = seed + 12345                   model noise counter = the learner's call counter (incremented before the call)
Test infrastructure only (imports oracle/)."""
import numpy as np
import torch

from oracle import mpg_oracle as O


class OracleConfig2Loop(object):
    def __init__(self, flat_by_name, seed=0, num_agent=64, batch_size=64, replay_batch_size=256, replay_starts=512, capacity=500000,
                 sampling_interval=1, explore_sigma=0.1, dtype=torch.float32, alg='MPG-v2', num_future_data=0):
        # TD3 (learners/td3.py:150-188, uniform replay): the same loop, another learner.  MPG-v1 (networks [Q1 | policy]): the critic's
        # target is the 25-step REAL-env return from (s, a_replay) (mpg_learner.py:109-124,146-169), recomputed with a new minibatch every
        # num_batch_reuse = 10 gradient calls and kept in between (mpg_learner.py:402-403; train_script.py's default for v1)
        assert alg in ('MPG-v2', 'TD3', 'MPG-v1')
        self.alg = alg
        self.reuse = 10 if alg == 'MPG-v1' else 1
        K = self.K = num_future_data                      # look-ahead entries of the observation (path_tracking_env.py:385-402)
        # PathTracking-v0 defaults (select [0, 25], delay_update 2, smoothing .2 / .5, ...)
        self.cfg = O.Cfg() if K == 0 else O.Cfg(obs_dim=6 + K, obs_scale=list(O.OBS_SCALE_PT) + [1.] * K)
        self.names = ['Q1', 'policy'] if alg == 'MPG-v1' else ['Q1', 'Q2', 'policy']
        self.w = {k: np.array(flat_by_name[k], np.float32) for k in self.names}
        self.tgt = {k: v.copy() for k, v in self.w.items()}
        self.opt = {k: O.AdamState(v.size) for k, v in self.w.items()}
        self.dtype = dtype
        self.num_agent, self.sample_iters, self.B = num_agent, max(1, batch_size // num_agent), replay_batch_size
        self.sampling_interval, self.sigma = sampling_interval, explore_sigma
        self.w_seed = seed * 1000003
        self.rb_seed = seed * 7919
        self.l_seed = seed + 12345
        self.noise_ctr = self.env_ctr = self.replay_times = self.counter = 0
        self.env = O.PathTrackingEnvOracle(num_agent, K)
        self._redraw(np.ones(num_agent, bool))            # OffPolicyWorker.__init__: env.reset()
        self.cap = capacity
        self.ring = dict(obs=np.zeros((capacity, 6 + K), np.float32), act=np.zeros((capacity, 2), np.float32), rew=np.zeros(capacity, np.float32),
                         obs2=np.zeros((capacity, 6 + K), np.float32), done=np.zeros(capacity, np.float32))
        self.size = self.next = 0
        self.iteration = 0
        self.stats = None
        while self.size < replay_starts:                  # optimizer.py:310-313
            self.sample()

    def _redraw(self, mask):
        """env.reset(): agents with done == 1 get a fresh state from the reset law (path_tracking_env.py:423-454) - here the
        device's Philox stream (agent, reset counter)"""
        full, _ = O.reset_law_philox(self.num_agent, self.w_seed, self.env_ctr)
        self.env_ctr += 1
        e = self.env
        e.veh_full_state = full if e.veh_full_state is None else np.where(mask[:, None], full, e.veh_full_state).astype(np.float32)
        e.veh_state = e.veh_full_state.copy()
        x = e.veh_full_state[:, -1]
        e.veh_state[:, 4] = e.veh_full_state[:, 4] - O.path_phi(x)
        e.veh_state[:, 3] = e.veh_full_state[:, 3] - O.path_y(x)
        e.obs = e._get_obs(e.veh_state, e.veh_full_state)

    def nets(self):
        return O.Nets(self.cfg, self.w, flat_targets=self.tgt, dtype=self.dtype)

    def sample(self):
        """OffPolicyWorker.sample, worker.py:91-119"""
        nets = self.nets()
        for _ in range(self.sample_iters):
            obs = self.env.obs.astype(np.float32).copy()
            with torch.no_grad():
                a = nets.compute_action(O.process_obses(self.cfg, torch.as_tensor(obs).to(self.dtype))).numpy().astype(np.float32)
            a = (a + O.explore_noise_philox(self.num_agent, 2, self.sigma, self.w_seed, self.noise_ctr)).astype(np.float32)
            self.noise_ctr += 1
            obs2, rew, done, _ = self.env.step(a)
            sl = (self.next + np.arange(self.num_agent)) % self.cap
            r = self.ring
            r['obs'][sl], r['act'][sl], r['rew'][sl], r['obs2'][sl], r['done'][sl] = obs, a, rew, obs2, np.asarray(done, np.float32)
            self.next = (self.next + self.num_agent) % self.cap
            self.size = min(self.size + self.num_agent, self.cap)
            self._redraw(np.asarray(done, bool))

    def step(self):
        """SingleProcessOffPolicyOptimizer.step, optimizer.py:330-362"""
        it = self.iteration
        if it % self.sampling_interval == 0:
            self.sample()
        self.replay_times += 1                            # replay() is called every iteration (optimizer.py:340-341) ...
        r = self.ring
        nets = self.nets()
        if self.counter % self.reuse == 0:                # ... but the learner takes a new minibatch only every `reuse` calls
            idx = self.idx = O.uniform_indices_philox(self.size, self.B, self.rb_seed, self.replay_times)
            self.batch = [r['obs'][idx], r['act'][idx], r['rew'][idx], r['obs2'][idx], r['done'][idx]]
            if self.alg == 'MPG-v1':
                self.targets = O.n_step_target(self.cfg, nets, self.batch[0], self.batch[1])
        self.counter += 1
        batch = self.batch
        if self.alg == 'TD3':       # target-policy smoothing noise (td3.py:74): mpg_normal_fill(learner seed, call counter)
            eps = O.normal_fill_philox(self.B * 2, self.l_seed, self.counter).reshape(self.B, 2)
            grads, st = O.td3_compute_gradient(self.cfg, nets, batch, eps)
        elif self.alg == 'MPG-v1':
            eps = O.model_noise_philox(self.cfg.n, self.B, self.l_seed, self.counter)
            grads, st = self._v1_gradient(nets, batch, self.targets, eps, it)
        else:
            eps = O.model_noise_philox(self.cfg.n, self.B, self.l_seed, self.counter)
            grads, st = O.mpg_compute_gradient(self.cfg, nets, batch, eps, it, 'MPG-v2')
        g, o = {}, 0
        for k in self.names:
            g[k] = np.concatenate([x.ravel() for x in grads[o:o + 6]]).astype(np.float32)
            o += 6
        O.apply_gradients(self.cfg, self.w, self.tgt, self.opt, g, it, self.names)
        self.stats = st
        self.iteration += 1

    # ---- data-parallel form: R replicas with their own worker / replay / noise streams and SHARED parameters ----
    def draw(self):
        """this rank's part of an iteration up to the gradient: sample (every sampling_interval-th iteration), replay, model noise"""
        assert self.alg == 'MPG-v2'
        if self.iteration % self.sampling_interval == 0:
            self.sample()
        self.replay_times += 1
        idx = self.idx = O.uniform_indices_philox(self.size, self.B, self.rb_seed, self.replay_times)
        self.counter += 1
        r = self.ring
        return [r['obs'][idx], r['act'][idx], r['rew'][idx], r['obs2'][idx], r['done'][idx]], \
            O.model_noise_philox(self.cfg.n, self.B, self.l_seed, self.counter)

    def _v1_gradient(self, nets, batch, targets, eps, iteration):
        """MPGLearner.compute_gradient for MPG-v1 with the CACHED n-step targets (mpg_learner.py:401-455): O.mpg_compute_gradient with the
        target taken from the caller instead of being recomputed"""
        cfg, dt = self.cfg, nets.dtype
        obs, act = [torch.as_tensor(np.asarray(b, np.float32)).to(dt) for b in batch[:2]]
        q_losses, q_grads = O.q_forward_and_backward(cfg, nets, obs, act, torch.as_tensor(targets).to(dt), ['Q1'])
        qg, qn = O.clip_by_global_norm(q_grads[0], cfg.clip)
        reduced, _, _ = O.model_rollout_for_policy_update(cfg, nets, obs, torch.as_tensor(eps).to(dt))
        ws = O.rule_based_weights(iteration, cfg.total_ite, cfg.eta, cfg.select, dt)
        total_loss = torch.sum(ws.detach() * torch.stack([-reduced[k] for k in cfg.select]))
        pg, pn = O.clip_by_global_norm(list(torch.autograd.grad(total_loss, nets.w['policy'])), cfg.clip)
        st = dict(q_loss1=q_losses[0].numpy(), value_mean=reduced[0].detach().numpy(), targets=np.asarray(targets))
        return [g.detach().numpy() for g in qg + pg], st

    def flat(self):
        return np.concatenate([self.w[k] for k in self.names]), np.concatenate([self.tgt[k] for k in self.names])


def data_parallel_step(loops):
    """One iteration of R replicas (OracleConfig2Loop objects built from the SAME weights with seeds 0 .. R-1): the gradient the ranks
    exchange is the sum of their 1 / B_global-scaled partials = the gradient of the concatenated minibatch (every loss is a mean over
    the batch); the clip runs on that sum, Adam / Polyak identically on every replica (the reference's own multi-learner
    form applies stale gradients one at a time, optimizer.py:60-94 - this is the synchronous statement of it)."""
    lead = loops[0]
    it = lead.iteration
    for lp in loops[1:]:                                   # replicas share ONE set of parameters, targets and optimizer states
        lp.w, lp.tgt, lp.opt = lead.w, lead.tgt, lead.opt
    parts = [lp.draw() for lp in loops]
    batch = [np.concatenate([p[0][k] for p in parts], 0) for k in range(5)]
    eps = np.concatenate([p[1] for p in parts], 1)
    grads, st = O.mpg_compute_gradient(lead.cfg, lead.nets(), batch, eps, it, 'MPG-v2')
    g, o = {}, 0
    for k in lead.names:
        g[k] = np.concatenate([x.ravel() for x in grads[o:o + 6]]).astype(np.float32)
        o += 6
    O.apply_gradients(lead.cfg, lead.w, lead.tgt, lead.opt, g, it, lead.names)
    for lp in loops:
        lp.stats = st
        lp.iteration += 1


class OracleConfig4Loop(OracleConfig2Loop):
    """Config 4's loop - TD3Learner + PrioritizedReplayBuffer (buffer.py:94-189; optimizer.py:351-353: update_priorities between
    compute_gradient and apply_gradients) - as a TEACHER-FORCED oracle run: proportional sampling is discontinuous in the priorities (a
    1e-6 relative difference in one float32 |td| moves the float64 prefix sums enough to flip ~1 drawn index in 256), so the loop takes
    each iteration's indices from the device and checks, against ITS OWN trees, that every one of them is a find_prefixsum_idx answer
    within the band its leaves' distance to the device's allows.  Everything else - trees (float64, the reference's association), IS
    weights, |td| + eps priorities, max_priority, new transitions entering at max priority, the TD3 gradients, Adam - it computes itself.
    The canonical PER of mpg_amd/buffer.py (the shipped constructor is dead code, SURVEY B-3): alpha 0.6, beta 0.4, eps 1e-6."""

    def __init__(self, flat_by_name, alpha=0.6, beta=0.4, eps=1e-6, **kw):
        cap = 1
        while cap < kw.get('capacity', 500000):
            cap *= 2
        self.tree_cap, self.alpha, self.beta, self.per_eps = cap, alpha, beta, eps
        self.leaves = np.zeros(cap, np.float64)                       # sum-tree leaves (the min tree holds the same values, inf where unset)
        self.max_priority = 1.0                                       # buffer.py:125 (a python float: float64)
        self._seen = np.zeros(cap, bool)
        super().__init__(flat_by_name, alg='TD3', **kw)

    def sample(self):
        first, n0 = self.next, self.size
        super().sample()
        k = self.sample_iters * self.num_agent
        sl = (first + np.arange(k)) % self.cap
        self.leaves[sl] = float(self.max_priority) ** self.alpha      # buffer.py:133-136: weight = max priority
        self._seen[sl] = True

    def trees(self):
        st = O.heap_tree(self.leaves, np.add)
        mt = O.heap_tree(np.where(self._seen, self.leaves, np.inf), np.minimum)
        return st, mt

    def is_weights(self, st, mt, idx):
        """sample_with_weights_and_idxes, buffer.py:146-158"""
        total = st[1]
        p_min = mt[1] / total
        max_w = (p_min * self.size) ** (-self.beta)
        return ((st[self.tree_cap + idx] / total * self.size) ** (-self.beta) / max_w)

    def step(self, forced_idx):
        it = self.iteration
        if it % self.sampling_interval == 0:
            self.sample()
        self.replay_times += 1
        st, mt = self.trees()
        self.u = O.per_uniform_philox(self.B, self.rb_seed, self.replay_times)
        self.own_idx = O.find_prefixsum_idx_batch(st, self.u * st[1])        # what this loop's own trees would have drawn
        self.sum_tree = st
        idx = self.idx = np.asarray(forced_idx, np.int64)
        self.weights = self.is_weights(st, mt, idx)
        self.min_leaf = mt[1]
        r = self.ring
        batch = self.batch = [r['obs'][idx], r['act'][idx], r['rew'][idx], r['obs2'][idx], r['done'][idx]]
        self.counter += 1
        nets = self.nets()
        eps = O.normal_fill_philox(self.B * 2, self.l_seed, self.counter).reshape(self.B, 2)
        grads, stt = O.td3_compute_gradient(self.cfg, nets, batch, eps)
        with torch.no_grad():
            td = O.td_error(self.cfg, nets, *[torch.as_tensor(b).to(self.dtype) for b in batch[:4]])
        self.td = np.asarray(td, np.float32)
        # update_priorities, buffer.py:166-189 (sequential: the LAST occurrence of a duplicate index wins); priority = |td| + eps
        p = np.abs(self.td.astype(np.float64)) + self.per_eps
        last = {}
        for k, j in enumerate(idx):
            last[int(j)] = k
        jj = np.fromiter(last.keys(), np.int64)
        kk = np.fromiter(last.values(), np.int64)
        self.leaves[jj] = p[kk] ** self.alpha
        self.max_priority = max(self.max_priority, float(p.max()))
        g, o = {}, 0
        for k in self.names:
            g[k] = np.concatenate([x.ravel() for x in grads[o:o + 6]]).astype(np.float32)
            o += 6
        O.apply_gradients(self.cfg, self.w, self.tgt, self.opt, g, it, self.names)
        self.stats = stt
        self.iteration += 1
