"""Device-resident vectorised environment - mirror of PathTrackingEnv
(envs_and_models/path_tracking_env.py:356-487): same method names, argument meaning and quirks
(done is always True); arrays are torch tensors on the GPU instead of numpy."""
import torch

from . import _lib as L

ENV_KIND = {'PathTracking-v0': 0, 'InvertedPendulumConti-v0': 1}


class PathTrackingEnv(object):
    """gym-style vector env, `num_agent` agents stepped by one HIP launch (mpg_env_step)."""
    obs_dim, act_dim = 6, 2

    def __init__(self, num_future_data=0, num_agent=1, device='cuda', seed=0, **kwargs):
        if num_future_data != 0:
            raise NotImplementedError('num_future_data > 0 is outside the hot-path scope (SURVEY.md §8 f3)')
        self.num_agent = num_agent
        self.device = torch.device(device)
        self.seed, self._ctr = int(seed), 0
        n = num_agent
        self._state = torch.zeros(8, n, dtype=torch.float32, device=self.device)
        self.obs = torch.zeros(n, 6, dtype=torch.float32, device=self.device)
        self.reward = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.done = torch.ones(n, dtype=torch.uint8, device=self.device)
        self.done_intended = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self._initialised = False

    # veh_full_state / veh_state views in the reference's column order (for tests and evaluators)
    @property
    def veh_full_state(self):
        return self._state[:6].t().contiguous()

    @property
    def veh_state(self):
        s = self._state
        return torch.stack([s[0], s[1], s[2], s[6], s[7], s[5]], 1)

    def reset(self, **kwargs):
        """reset(init_obs=obs) rebuilds the state from obs (:411-421); reset() re-draws agents with done==1
        (:423-454) - every agent on the first call."""
        if 'init_obs' in kwargs:
            init_obs = kwargs['init_obs'].to(self.device, torch.float32).contiguous()
            assert init_obs.shape == (self.num_agent, 6)
            L.call('mpg_env_reset_from_obs', L.c_int(0), L.c_int(self.num_agent), L.ptr(self._state),
                   L.ptr(init_obs), L.stream())
            self.obs = init_obs
            self._initialised = True
            return self.obs
        mask = self.done if self._initialised else None
        obs = torch.empty_like(self.obs)        # never write into a tensor already handed to the caller
        L.call('mpg_env_reset', L.c_int(0), L.c_int(self.num_agent), L.ptr(self._state), L.ptr(mask),
               L.c_u64(self.seed), L.c_u64(self._ctr), L.ptr(obs), L.stream())
        self._ctr += 1
        self._initialised = True
        self.obs = obs
        return self.obs

    def step(self, action):
        """action [num_agent, 2] in [-1, 1] -> (obs, reward, done, info) (:456-472)."""
        if not (action.dtype == torch.float32 and action.device == self.device and action.is_contiguous()):
            action = action.to(self.device, torch.float32).contiguous()
        assert action.shape == (self.num_agent, 2)
        obs = torch.empty_like(self.obs)
        reward = torch.empty_like(self.reward)
        done = torch.empty_like(self.done)      # fresh outputs: callers keep them (replay batches)
        L.call('mpg_env_step', L.c_int(0), L.c_int(self.num_agent), L.ptr(self._state), L.ptr(action),
               L.ptr(obs), L.ptr(reward), L.ptr(done), L.ptr(self.done_intended), L.stream())
        self.obs, self.reward, self.done = obs, reward, done
        return self.obs, self.reward, self.done, {}
