"""Stand-in for `tensorflow_probability` (only `distributions.Normal(...).sample()` is on the path:
path_tracking_env.py:119, inverted_pendulum_model.py:61).  TEST INFRASTRUCTURE ONLY."""
import types

import tensorflow as tf


class Normal(object):
    def __init__(self, loc, scale, **kw):
        self.loc, self.scale = tf.convert_to_tensor(loc), scale

    def sample(self, *a, **k):
        return self.loc + self.scale * tf._std_normal(self.loc.shape)


class _Any(object):
    def __getattr__(self, k):
        raise NotImplementedError('tfp.%s is outside the oracle surface (SAC only)' % k)


distributions = types.SimpleNamespace(Normal=Normal)
bijectors = _Any()
