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
    """tf.keras.optimizers.schedules.PolynomialDecay (cycle=False) as TensorFlow evaluates it: every operand is cast to the dtype of
    `initial_learning_rate` (float32; the float64 yard-stick run uses float64 throughout),
        step' = min(step, decay_steps);  p = step' / decay_steps;  lr = (lr0 - lr_end) * (1 - p) ** power + lr_end.
    Call site: policy.py:54,62 (`PolynomialDecay(*policy_lr_schedule)` = (lr0, decay_steps, lr_end))."""

    def __init__(self, initial_learning_rate, decay_steps, end_learning_rate=0.0001, power=1.0, cycle=False, name=None):
        assert not cycle
        self.initial_learning_rate, self.decay_steps = initial_learning_rate, decay_steps
        self.end_learning_rate, self.power = end_learning_rate, power

    def __call__(self, step):
        dt = tf.REF_DTYPE
        c = lambda x: torch.as_tensor(float(x), dtype=dt)
        lr0, lr_end, power, S = c(self.initial_learning_rate), c(self.end_learning_rate), c(self.power), c(self.decay_steps)
        p = torch.minimum(c(int(step)), S) / S
        return (lr0 - lr_end) * torch.pow(c(1.0) - p, power) + lr_end


class Adam(object):
    """tf.keras.optimizers.Adam (OptimizerV2, amsgrad=False) restated from its published update - the `ApplyAdam` kernel the dense
    path calls (`training_ops.resource_apply_adam`), all in the variable's dtype:
        lr_t    = learning_rate(iterations)                       (the schedule sees the count BEFORE this call's increment)
        t       = iterations + 1;  b1p = beta_1 ** t;  b2p = beta_2 ** t
        alpha   = lr_t * sqrt(1 - b2p) / (1 - b1p)
        m      += (g - m) * (1 - beta_1);   v += (g * g - v) * (1 - beta_2);   var -= (m * alpha) / (sqrt(v) + epsilon)
    `epsilon` (1e-7) sits OUTSIDE the bias-corrected root; `iterations` is per optimizer object and moves once per
    `apply_gradients` call.  TensorFlow itself is absent from this image, so the ARITHMETIC is this restatement (stated as such in
    DESIGN.md section 2); what the stand-in buys is that the reference's own control flow around it (policy.py:123-171: which
    optimizer steps when, per-optimizer counters, delayed policy / target updates) executes unmodified."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False, name='Adam', **kw):
        assert not amsgrad
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon
        self._name = name
        self.iterations = 0
        self._slots = {}                                   # id(variable) -> (m, v)

    def get_slot(self, var, name):
        return self._slots[id(var)][0 if name == 'm' else 1]

    def apply_gradients(self, grads_and_vars, name=None):
        with torch.no_grad():
            for g, var in grads_and_vars:
                dt = var.dtype
                c = lambda x, dt=dt: torch.as_tensor(float(x), dtype=dt)
                lr = self.learning_rate(self.iterations) if callable(self.learning_rate) else self.learning_rate
                lr_t = torch.as_tensor(lr).to(dt)
                t = c(self.iterations + 1)
                b1, b2, eps = c(self.beta_1), c(self.beta_2), c(self.epsilon)
                b1p, b2p = torch.pow(b1, t), torch.pow(b2, t)
                alpha = lr_t * torch.sqrt(c(1.0) - b2p) / (c(1.0) - b1p)
                if id(var) not in self._slots:
                    z = torch.Tensor.detach(var).as_subclass(torch.Tensor)
                    self._slots[id(var)] = (torch.zeros_like(z), torch.zeros_like(z))
                m, v = self._slots[id(var)]
                g = torch.as_tensor(np.asarray(g)) if not isinstance(g, torch.Tensor) else g
                g = g.detach().as_subclass(torch.Tensor).to(dt)
                m += (g - m) * (c(1.0) - b1)
                v += (g * g - v) * (c(1.0) - b2)
                var.as_subclass(torch.Tensor).sub_((m * alpha) / (torch.sqrt(v) + eps))
        self.iterations += 1


layers = _ns('layers', Dense=Dense)
initializers = _ns('initializers', Orthogonal=Orthogonal, Constant=Constant, Zeros=Zeros)
schedules = _ns('optimizers.schedules', PolynomialDecay=PolynomialDecay)
optimizers = _ns('optimizers', Adam=Adam, schedules=schedules)
