"""OffPolicyWorker - device mirror of worker.py:25-123: the env-step producer that also owns the master weights and
the optimizer state (`apply_gradients`).  `sample()` keeps the reference's loop (policy -> + N(0, sigma) ->
env.step -> env.reset, worker.py:91-119) but every stage is one HIP launch over all `num_agent` agents."""
import torch

from .envs import make_env
from . import ops


class OffPolicyWorker(object):
    def __init__(self, policy_cls, env_id, args, worker_id, device='cuda'):
        self.worker_id = worker_id
        self.args = args
        self.device = torch.device(device)
        self.num_agent = int(args.num_agent)
        seed = int(getattr(args, 'seed', 0)) * 1000003 + int(worker_id)
        # PathTracking-v0, or InvertedPendulumConti-v0 (the reference wraps the single MuJoCo env in DummyVecEnv with
        # num_agent 1, train_script4mujoco.py:328; here `num_agent` pendulums step in one launch)
        self.env = make_env(env_id, num_agent=self.num_agent, num_future_data=getattr(args, 'num_future_data', 0), device=device, seed=seed)
        self.policy_with_value = policy_cls(**vars(args), device=device)
        self.batch_size = int(args.batch_size)
        self.obs = self.env.reset()
        self.explore_sigma = args.explore_sigma
        self.seed = seed
        self.iteration = 0
        self.num_sample = 0
        self.sample_times = 0
        self.nan_check_interval = int(getattr(args, 'nan_check_interval', 100))
        self._noise_ctr = 0
        self.stats = {}

    def get_stats(self):
        self.stats.update(dict(worker_id=self.worker_id, num_sample=self.num_sample))
        return self.stats

    def get_weights(self):
        return self.policy_with_value.get_weights()

    def set_weights(self, weights):
        return self.policy_with_value.set_weights(weights)

    def save_weights(self, save_dir, iteration):
        self.policy_with_value.save_weights(save_dir, iteration)

    def load_weights(self, load_dir, iteration):
        self.policy_with_value.load_weights(load_dir, iteration)

    def apply_gradients(self, iteration, grads):
        self.iteration = iteration
        self.policy_with_value.apply_gradients(iteration, grads)

    def get_ppc_params(self):
        return {}

    def set_ppc_params(self, params):
        pass

    def sample(self):
        """-> (obs, act, RAW reward, obs', done) stacked over batch_size/num_agent env steps (worker.py:91-119)."""
        pw = self.policy_with_value
        out = [[], [], [], [], []]
        iters = max(1, self.batch_size // self.num_agent)
        for _ in range(iters):
            obs = self.obs
            action = ops.policy_action(pw.cfg, pw.net('policy'), obs, explore_sigma=float(self.explore_sigma or 0.),
                                       seed=self.seed, ctr=self._noise_ctr)
            self._noise_ctr += 1
            obs_tp1, reward, done, _ = self.env.step(action)          # fresh tensors every call (never aliased later)
            for lst, x in zip(out, (obs, action, reward, obs_tp1, done)):
                lst.append(x)
            self.obs = self.env.reset()          # PathTracking: done is always 1, every agent is re-drawn; the pendulum
                                                 # re-draws the agents that fell (inverted_pendulum_conti.py:17-18)
        batch = tuple(x[0] for x in out) if iters == 1 else tuple(torch.cat(x, 0) for x in out)
        self.num_sample += batch[0].shape[0]
        self.sample_times += 1
        # judge_is_nan (worker.py:95-107) runs inside the policy kernel: a NaN observation or action sets MPG_STATUS_NAN in the
        # policy's status word; it is read (one host synchronisation) every `nan_check_interval` calls, not per step
        if self.sample_times % self.nan_check_interval == 0:
            self.policy_with_value.check_status()
        return batch

    def sample_with_count(self):
        batch = self.sample()
        return batch, batch[0].shape[0]
