"""Evaluator - device mirror of evaluator.py:25-236: deterministic fixed-length parallel episodes on the
HIP env and the reference's per-episode metrics (evaluator.py:160-211), averaged over the agents.  Not on the throughput
path; it is the learning-curve check that the fast path still learns (the reference's plots use a base score of -30,
ploter.py:85)."""
import torch

from . import ops
from .envs import make_env


class Evaluator(object):
    def __init__(self, policy_cls, env_id, args, device='cuda'):
        self.env_id = env_id
        self.args = args
        self.device = torch.device(device)
        self.num_eval_agent = int(getattr(args, 'num_eval_agent', 5))
        self.fixed_steps = int(getattr(args, 'fixed_steps', 200))
        self.env = make_env(env_id, num_agent=self.num_eval_agent, num_future_data=getattr(args, 'num_future_data', 0), device=device,
                            seed=int(getattr(args, 'seed', 0)) + 424242)
        self.policy_with_value = policy_cls(**vars(args), device=device)
        self.iteration = 0
        self.stats = {}

    def set_weights(self, weights):
        if weights is self.policy_with_value:
            return
        self.policy_with_value.set_weights(weights)

    def share_policy(self, policy):
        self.policy_with_value = policy

    def set_ppc_params(self, params):
        pass

    def run_n_episodes_parallel(self, n=None, init_obs=None):
        """evaluator.py:118-156: reset, `fixed_steps` deterministic steps (done is ignored), metrics per episode.
        init_obs (optional, [num_eval_agent, obs_dim]): start from these observations instead of a fresh reset() draw
        (the parity test starts from the states the reference's own env drew)."""
        pw = self.policy_with_value
        if init_obs is not None:
            obses = self.env.reset(init_obs=init_obs)
        else:
            self.env._initialised = False                  # fresh draw of every agent
            obses = self.env.reset()
        obs_l, act_l, rew_l = [], [], []
        for _ in range(self.fixed_steps):
            actions = ops.policy_action(pw.cfg, pw.net('policy'), obses)      # compute_mode, policy.py:173-177
            obs_l.append(obses)
            act_l.append(actions)
            obses, rewards, _, _ = self.env.step(actions)
            rew_l.append(rewards)
        # the reference accumulates in numpy: np.mean / python sum() promote to float64 (evaluator.py:141-142,172-178)
        obs, act, rew = torch.stack(obs_l).double(), torch.stack(act_l).double(), torch.stack(rew_l).double()   # [T, N, .]
        rms = lambda x: torch.sqrt(torch.mean(torch.square(x), 0))
        per_episode = dict(episode_return=rew.sum(0), episode_len=torch.full_like(rew[0], float(self.fixed_steps)))
        if self.env_id == 'PathTracking-v0':               # metrics_for_an_episode, evaluator.py:160-184
            per_episode.update(
                delta_y_mse=rms(obs[:, :, 3]), delta_phi_mse=rms(obs[:, :, 4]), delta_v_mse=rms(obs[:, :, 0]),
                stationary_rew_mean=rew[20:].mean(0), steer_mse=rms(act[:, :, 0] * (1.2 * 3.141592653589793 / 9)),
                acc_mse=rms(act[:, :, 1] * 3.))
        else:                                              # InvertedPendulumConti-v0, evaluator.py:185-211
            for j, nm in enumerate(('x', 'theta', 'xdot', 'thetadot')):
                per_episode[nm + '_mean'] = obs[:, :, j].mean(0)
                per_episode[nm + '_var'] = obs[:, :, j].var(0, unbiased=False)
                per_episode[nm + '_mse'] = rms(obs[:, :, j])
                per_episode[nm + '_mse_25'] = rms(obs[:25, :, j])
        mean = {k: float(v.mean().item()) for k, v in per_episode.items()}
        pw.check_status()          # NaN observations / actions, the engine's envelope: an evaluation on invalid numbers raises
        return per_episode, mean

    def run_evaluation(self, iteration):
        self.iteration = iteration
        _, mean = self.run_n_episodes_parallel()
        self.stats = dict(iteration=iteration, **mean)
        return mean
