"""Stand-in for the slice of `tensorflow.keras` used by the reference (model.py:20-43, policy.py:54-70).

TEST INFRASTRUCTURE ONLY (see ../__init__.py).  Dense kernels are stored (in, out) like Keras.
Initialiser RNG streams are NOT those of Keras - every golden carries its weights as inputs.
"""
import sys
import types

import numpy as np
import torch

import tensorflow as tf


def _ns(name, **kw):
    m = types.ModuleType(__name__ + '.' + name)
    m.__dict__.update(kw)
    sys.modules[m.__name__] = m
    return m


class Orthogonal(object):
    def __init__(self, gain=1.0, seed=None):
        self.gain = float(gain)

    def __call__(self, shape):
        rows, cols = shape
        a = torch.randn(max(rows, cols), min(rows, cols), dtype=torch.float64)
        q, r = torch.linalg.qr(a)
        q = q * torch.sign(torch.diagonal(r))
        if rows < cols:
            q = q.T
        return (self.gain * q[:rows, :cols]).to(tf.REF_DTYPE)


class Constant(object):
    def __init__(self, value=0.0):
        self.value = value

    def __call__(self, shape):
        return torch.full(tuple(shape), float(self.value), dtype=tf.REF_DTYPE)


class Zeros(Constant):
    pass


_ACT = {None: lambda x: x, 'linear': lambda x: x, 'tanh': torch.tanh,
        'elu': torch.nn.functional.elu, 'relu': torch.relu}


class Dense(object):
    def __init__(self, units, activation=None, kernel_initializer=None, bias_initializer=None,
                 dtype=None, name=None, **kw):
        self.units = units
        self.activation = _ACT[activation]
        self.kernel_initializer = kernel_initializer or Orthogonal(1.0)
        self.bias_initializer = bias_initializer or Constant(0.0)
        self.kernel = self.bias = None

    def build(self, in_dim):
        self.kernel = tf.Variable(self.kernel_initializer((in_dim, self.units)))
        self.bias = tf.Variable(self.bias_initializer((self.units,)))
        return self.units

    @property
    def trainable_weights(self):
        return [self.kernel, self.bias]

    def __call__(self, x):
        x = tf.convert_to_tensor(x)
        if x.dtype != self.kernel.dtype:
            x = x.to(self.kernel.dtype)
        return self.activation(torch.matmul(x, self.kernel) + self.bias)


class Sequential(object):
    def __init__(self, layers=None, name=None):
        self.layers = list(layers or [])

    def build(self, in_dim):
        for l in self.layers:
            in_dim = l.build(in_dim)
        return in_dim

    @property
    def trainable_weights(self):
        return [w for l in self.layers for w in l.trainable_weights]

    def __call__(self, x):
        for l in self.layers:
            x = l(x)
        return x


class Model(object):
    def __init__(self, name=None, **kw):
        object.__setattr__(self, '_layers', [])
        self.name = name

    def __setattr__(self, k, v):
        if isinstance(v, (Dense, Sequential)):
            self._layers.append(v)
        object.__setattr__(self, k, v)

    def build(self, input_shape):
        d = input_shape[-1]
        for l in self._layers:
            d = l.build(d)

    @property
    def trainable_weights(self):
        return [w for l in self._layers for w in l.trainable_weights]

    def get_weights(self):
        return [w.numpy().copy() for w in self.trainable_weights]

    def set_weights(self, weights):
        for w, v in zip(self.trainable_weights, weights):
            w.assign(v)

    def __call__(self, x, **kw):
        return self.call(x, **kw)


class PolynomialDecay(object):
    def __init__(self, initial_learning_rate, decay_steps, end_learning_rate=0.0001, power=1.0):
        self.args = (initial_learning_rate, decay_steps, end_learning_rate, power)


class Adam(object):
    """Not exercised by any golden (a20 is restated from the published TF formulae)."""

    def __init__(self, learning_rate=0.001, name='Adam', **kw):
        self.learning_rate = learning_rate
        self._name = name

    def apply_gradients(self, grads_and_vars):
        raise NotImplementedError('Keras Adam is not reproduced by the stand-in')


layers = _ns('layers', Dense=Dense)
initializers = _ns('initializers', Orthogonal=Orthogonal, Constant=Constant, Zeros=Zeros)
schedules = _ns('optimizers.schedules', PolynomialDecay=PolynomialDecay)
optimizers = _ns('optimizers', Adam=Adam, schedules=schedules)
