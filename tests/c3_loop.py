"""The config-3 training loop (train_scripts/train_script4mujoco.py:296-411: OffPolicyWorker -> ReplayBuffer -> NADPLearner ->
clip / Adam / Polyak in SingleProcessOffPolicyOptimizer.step's order, optimizer.py:330-362) as a pure ORACLE run that consumes
EXACTLY the random inputs the device loop consumes.  Every random input of the device loop is a counter-based Philox draw keyed by
(seed, call counter): the cart-pole reset law, the uniform replay indices and the model noise of the two NADP rollouts - all
restated host-side in oracle/mpg_oracle.py (cart_pole_reset_philox, uniform_indices_philox, model_noise_philox).  Started from
the device's initial weights this loop must therefore follow the device loop iteration by iteration.

Seeds and counters are the ones mpg_amd derives from `args.seed` (worker.py / buffer.py / learners.py of this repo):
    worker / env seed  = seed * 1000003 + worker_id      env reset counter: one per env.reset() call, first call in the constructor
    replay seed        = seed * 7919 + buffer_id         counter = replay_times (incremented before the draw)
    learner seed       = seed + 12345                    model noise counters 2 k (Q-target rollout), 2 k + 1 (policy rollout),
                                                         k = the learner's call counter (incremented before the call)
Test infrastructure only (imports oracle/)."""
import numpy as np
import torch

from oracle import mpg_oracle as O


class OracleConfig3Loop(object):
    def __init__(self, flat_q1, flat_policy, seed=0, num_agent=64, batch_size=512, replay_batch_size=512, replay_starts=3000,
                 capacity=500000, sampling_interval=10, n=25, dtype=torch.float32, env_dtype=np.float32, model_noise=True,
                 model_drift=True):
        self.cfg = O.Cfg(env='InvertedPendulumConti-v0', select=[n], delay_update=1)
        self.cfg.n = n
        self.names = ['Q1', 'policy']
        self.w = {'Q1': np.array(flat_q1, np.float32), 'policy': np.array(flat_policy, np.float32)}
        self.tgt = {k: v.copy() for k, v in self.w.items()}
        self.opt = {k: O.AdamState(v.size) for k, v in self.w.items()}
        self.dtype = dtype
        self.num_agent, self.sample_iters, self.B = num_agent, max(1, batch_size // num_agent), replay_batch_size
        self.sampling_interval, self.n = sampling_interval, n
        self.env_seed = seed * 1000003
        self.rb_seed = seed * 7919
        self.l_seed = seed + 12345
        self.env_ctr = self.replay_times = self.counter = 0
        self.model_noise, self.model_drift = model_noise, model_drift
        self.env = O.InvertedPendulumContiOracle(num_agent, dtype=env_dtype)
        self.env.reset(init_obs=O.cart_pole_reset_philox(num_agent, self.env_seed, self.env_ctr))     # OffPolicyWorker.__init__
        self.env_ctr += 1
        self.cap = capacity
        self.ring_obs = np.zeros((capacity, 4), np.float32)
        self.ring_act = np.zeros((capacity, 1), np.float32)
        self.size = self.next = 0
        self.iteration = 0
        self.stats = None
        while self.size < replay_starts:                   # optimizer.py:310-313
            self.sample()

    def nets(self):
        return O.Nets(self.cfg, self.w, flat_targets=self.tgt, dtype=self.dtype)

    def sample(self):
        """OffPolicyWorker.sample, worker.py:91-119 (explore_sigma None for this config): policy -> env.step -> env.reset of the
        agents that are done (inverted_pendulum_conti.py:17-18 behind DummyVecEnv)."""
        nets = self.nets()
        for _ in range(self.sample_iters):
            obs = self.env.state.astype(np.float32)
            with torch.no_grad():
                a = nets.compute_action(O.process_obses(self.cfg, torch.as_tensor(obs).to(self.dtype))).numpy().astype(np.float32)
            _, _, done, _ = self.env.step(a)
            sl = (self.next + np.arange(self.num_agent)) % self.cap
            self.ring_obs[sl], self.ring_act[sl] = obs, a
            self.next = (self.next + self.num_agent) % self.cap
            self.size = min(self.size + self.num_agent, self.cap)
            fresh = O.cart_pole_reset_philox(self.num_agent, self.env_seed, self.env_ctr)
            self.env_ctr += 1
            self.env.state = np.where(done[:, None], fresh.astype(self.env.state.dtype), self.env.state)

    def noise(self, ctr):
        if not self.model_noise:           # controls of VERDICT r4: no noise (eps = 0), or no noise and no drift (0.1 + 0.5 eps = 0)
            return np.full((self.n, self.B), 0.0 if self.model_drift else -0.2, np.float32)
        return O.model_noise_philox(self.n, self.B, self.l_seed, ctr)

    def step(self):
        """SingleProcessOffPolicyOptimizer.step, optimizer.py:330-362"""
        it = self.iteration
        if it % self.sampling_interval == 0:
            self.sample()
        self.replay_times += 1
        idx = O.uniform_indices_philox(self.size, self.B, self.rb_seed, self.replay_times)
        self.idx = idx
        self.counter += 1
        eps_q, eps_pi = self.noise(2 * self.counter), self.noise(2 * self.counter + 1)
        grads, st = O.nadp_compute_gradient(self.cfg, self.nets(), [self.ring_obs[idx], self.ring_act[idx]], eps_q, eps_pi)
        g = {'Q1': np.concatenate([x.ravel() for x in grads[:6]]).astype(np.float32),
             'policy': np.concatenate([x.ravel() for x in grads[6:]]).astype(np.float32)}
        O.apply_gradients(self.cfg, self.w, self.tgt, self.opt, g, it, self.names)
        self.stats = st
        self.iteration += 1

    def flat(self):
        return np.concatenate([self.w[k] for k in self.names]), np.concatenate([self.tgt[k] for k in self.names])

    def evaluate(self, n_agent=16, steps=100, seed=1000):
        """deterministic 100-step episodes from the reset law (evaluator.py:124-211 for this env: no early termination inside
        run_n_episodes_parallel's fixed-step loop)"""
        env = O.InvertedPendulumContiOracle(n_agent)
        obs = env.reset(init_obs=O.cart_pole_reset_philox(n_agent, seed, 0))
        nets = self.nets()
        ret, th2 = np.zeros(n_agent), np.zeros(n_agent)
        for _ in range(steps):
            with torch.no_grad():
                a = nets.compute_action(O.process_obses(self.cfg, torch.as_tensor(obs).to(self.dtype))).numpy()
            obs, rew, _, _ = env.step(a)
            ret += rew
            th2 += obs[:, 1] ** 2
        return float(ret.mean()), float(np.sqrt(th2 / steps).mean())
