"""Stand-in for the few `gym` names the reference imports (path_tracking_env.py:13,356-379;
learners/mpg_learner.py:34).  TEST INFRASTRUCTURE ONLY."""
import sys
import types

import numpy as np


class Env(object):
    pass


class Box(object):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low = np.asarray(low).astype(dtype)
        self.high = np.asarray(high).astype(dtype)
        self.dtype = dtype
        self.shape = self.low.shape


spaces = types.ModuleType('gym.spaces')
spaces.Box = Box
sys.modules['gym.spaces'] = spaces


class Wrapper(Env):
    def __init__(self, env):
        self.env = env


core = types.ModuleType('gym.core')
core.Wrapper = Wrapper
core.Env = Env
sys.modules['gym.core'] = core

_REGISTRY = {}


def register(id, entry):            # noqa: A002
    _REGISTRY[id] = entry


def make(id, **kwargs):             # noqa: A002
    if id == 'PathTracking-v0':
        from envs_and_models.path_tracking_env import PathTrackingEnv
        return PathTrackingEnv(**kwargs)
    return _REGISTRY[id](**kwargs)
