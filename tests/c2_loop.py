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
                 sampling_interval=1, explore_sigma=0.1, dtype=torch.float32, alg='MPG-v2'):
        assert alg in ('MPG-v2', 'TD3')                   # TD3 (learners/td3.py:150-188, uniform replay): the same loop, another learner
        self.alg = alg
        self.cfg = O.Cfg()                                # PathTracking-v0 defaults (select [0, 25], delay_update 2, smoothing .2 / .5, ...)
        self.names = ['Q1', 'Q2', 'policy']
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
        self.env = O.PathTrackingEnvOracle(num_agent)
        self._redraw(np.ones(num_agent, bool))            # OffPolicyWorker.__init__: env.reset()
        self.cap = capacity
        self.ring = dict(obs=np.zeros((capacity, 6), np.float32), act=np.zeros((capacity, 2), np.float32), rew=np.zeros(capacity, np.float32),
                         obs2=np.zeros((capacity, 6), np.float32), done=np.zeros(capacity, np.float32))
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
        self.replay_times += 1
        idx = self.idx = O.uniform_indices_philox(self.size, self.B, self.rb_seed, self.replay_times)
        self.counter += 1
        r = self.ring
        batch = [r['obs'][idx], r['act'][idx], r['rew'][idx], r['obs2'][idx], r['done'][idx]]
        if self.alg == 'TD3':       # target-policy smoothing noise (td3.py:74): mpg_normal_fill(learner seed, call counter)
            eps = O.normal_fill_philox(self.B * 2, self.l_seed, self.counter).reshape(self.B, 2)
            grads, st = O.td3_compute_gradient(self.cfg, self.nets(), batch, eps)
        else:
            eps = O.model_noise_philox(self.cfg.n, self.B, self.l_seed, self.counter)
            grads, st = O.mpg_compute_gradient(self.cfg, self.nets(), batch, eps, it, 'MPG-v2')
        g, o = {}, 0
        for k in self.names:
            g[k] = np.concatenate([x.ravel() for x in grads[o:o + 6]]).astype(np.float32)
            o += 6
        O.apply_gradients(self.cfg, self.w, self.tgt, self.opt, g, it, self.names)
        self.stats = st
        self.iteration += 1

    def flat(self):
        return np.concatenate([self.w[k] for k in self.names]), np.concatenate([self.tgt[k] for k in self.names])
