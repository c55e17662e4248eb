"""Device-resident vectorised environments - mirrors of PathTrackingEnv
(envs_and_models/path_tracking_env.py:356-487) and of the vectorised InvertedPendulumContiEnv
(envs_and_models/inverted_pendulum_conti.py:5-30 behind utils/dummy_vec_env.py): same method names, argument meaning
and quirks (PathTracking's done is always True); arrays are torch tensors on the GPU instead of numpy."""
import torch

from . import _lib as L

ENV_KIND = {'PathTracking-v0': 0, 'InvertedPendulumConti-v0': 1}


class _DeviceVecEnv(object):
    """gym-style vector env: `num_agent` agents stepped by ONE HIP launch (mpg_env_step)."""
    kind, obs_dim, act_dim = 0, 6, 2

    def __init__(self, num_agent=1, device='cuda', seed=0):
        self.num_agent = num_agent
        self.device = torch.device(device)
        self.seed, self._ctr = int(seed), 0
        n = num_agent
        self._state = torch.zeros(8, n, dtype=torch.float32, device=self.device)       # MPG_ENV_STATE_DIM rows, opaque
        self.obs = torch.zeros(n, self.obs_dim, dtype=torch.float32, device=self.device)
        self.reward = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.done = torch.ones(n, dtype=torch.uint8, device=self.device)
        self.done_intended = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self._initialised = False

    def reset(self, **kwargs):
        """reset(init_obs=obs) rebuilds the state from obs (path_tracking_env.py:411-421); reset() re-draws agents with
        done==1 (:423-454) - every agent on the first call."""
        if 'init_obs' in kwargs:
            init_obs = kwargs['init_obs'].to(self.device, torch.float32).contiguous()
            assert init_obs.shape == (self.num_agent, self.obs_dim), (init_obs.shape, self.obs_dim)
            L.call('mpg_env_reset_from_obs', L.c_int(self.kind), L.c_int(self.num_agent), L.c_int(self.obs_dim),
                   L.ptr(self._state), L.ptr(init_obs), L.stream())
            self.obs = init_obs
            self._initialised = True
            return self.obs
        mask = self.done if self._initialised else None
        obs = torch.empty_like(self.obs)        # never write into a tensor already handed to the caller
        L.call('mpg_env_reset', L.c_int(self.kind), L.c_int(self.num_agent), L.c_int(self.obs_dim), L.ptr(self._state),
               L.ptr(mask), L.c_u64(self.seed), L.c_u64(self._ctr), L.ptr(obs), L.stream())
        self._ctr += 1
        self._initialised = True
        self.obs = obs
        return self.obs

    def step(self, action):
        """action [num_agent, act_dim] -> (obs, reward, done, info)."""
        if not (action.dtype == torch.float32 and action.device == self.device and action.is_contiguous()):
            action = action.to(self.device, torch.float32).contiguous()
        assert action.shape == (self.num_agent, self.act_dim)
        obs = torch.empty_like(self.obs)
        reward = torch.empty_like(self.reward)
        done = torch.empty_like(self.done)      # fresh outputs: callers keep them (replay batches)
        L.call('mpg_env_step', L.c_int(self.kind), L.c_int(self.num_agent), L.c_int(self.obs_dim), L.ptr(self._state),
               L.ptr(action), L.ptr(obs), L.ptr(reward), L.ptr(done), L.ptr(self.done_intended), L.stream())
        self.obs, self.reward, self.done = obs, reward, done
        return self.obs, self.reward, self.done, {}


class PathTrackingEnv(_DeviceVecEnv):
    """PathTrackingEnv(num_future_data, num_agent) - path_tracking_env.py:356-487.  Observations have 6 + num_future_data
    entries: the six base entries and the look-ahead delta-y terms of :385-402.  The env, worker, learner and evaluator
    serve 0 <= num_future_data <= 10 (policy inputs up to 16 wide, critic inputs up to 18: the 16- and 24-wide network kernels; the
    shipped parser default is 0, train_script.py:90; with look-ahead entries the learner takes the launch-per-stage path, the
    fused kernels are built for the base widths)."""
    kind, act_dim = 0, 2
    MAX_FUTURE = 10                            # MPG_ENV_MAX_FUTURE

    def __init__(self, num_future_data=0, num_agent=1, device='cuda', seed=0, **kwargs):
        assert 0 <= int(num_future_data) <= self.MAX_FUTURE, 'num_future_data in [0, %d]' % self.MAX_FUTURE
        self.num_future_data = int(num_future_data)
        self.obs_dim = 6 + self.num_future_data
        super().__init__(num_agent, device, seed)

    # veh_full_state / veh_state views in the reference's column order (for tests and evaluators)
    @property
    def veh_full_state(self):
        return self._state[:6].t().contiguous()

    @property
    def veh_state(self):
        s = self._state
        return torch.stack([s[0], s[1], s[2], s[6], s[7], s[5]], 1)


class InvertedPendulumContiEnv(_DeviceVecEnv):
    """The pendulum's REAL environment (the reference: MuJoCo behind DummyVecEnv, inverted_pendulum_conti.py:5-30) as an
    analytic RK4 cart-pole kernel (csrc/env_cart_pole.hip).  Parity with MuJoCo is unpinned (it cannot be installed here);
    the kernel is tested against the float64 restatement of the same equations in oracle/."""
    kind, obs_dim, act_dim = 1, 4, 1

    def __init__(self, num_agent=1, device='cuda', seed=0, **kwargs):
        super().__init__(num_agent, device, seed)


def make_env(env_id, num_agent=1, num_future_data=0, device='cuda', seed=0):
    """gym.make(env_id, ...) of the reference's worker / evaluator (worker.py:40-44, evaluator.py:33-36) for the two
    environments of the hot path."""
    if env_id == 'PathTracking-v0':
        return PathTrackingEnv(num_future_data=num_future_data, num_agent=num_agent, device=device, seed=seed)
    if env_id == 'InvertedPendulumConti-v0':
        return InvertedPendulumContiEnv(num_agent=num_agent, device=device, seed=seed)
    raise ValueError('no device environment for %r (PathTracking-v0, InvertedPendulumConti-v0)' % (env_id,))
