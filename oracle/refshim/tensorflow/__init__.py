"""Container-only stand-in for the `tensorflow` API surface the reference hot path touches.

TEST INFRASTRUCTURE ONLY.  TensorFlow is not installed in the build container (no network), so the
reference's own files (/root/reference/**/*.py) are imported *unmodified* on top of this torch-CPU
backed module by tests/golden/make_golden.py to produce the committed golden vectors.  Nothing in
mpg_amd/ imports this package; it never runs on the GPU box.

What is honoured:
  * `x += y` rebinds (TF tensors are immutable) - see RefTensor.__iadd__ & friends;
  * numpy arrays may meet tensors on either side of an operator;
  * `.numpy()` works on tensors that carry autograd history;
  * every random draw goes through `random.normal`, which can be fed from a recorded stream
    (`set_noise_source`) so the generator script controls the noise;
  * `REF_DTYPE` switches the whole graph between float32 and float64 (fp64 = error yard-stick).
"""
import contextlib
import sys
import types

import numpy as _np
import torch as _torch

_torch.set_num_threads(1)

REF_DTYPE = _torch.float32          # what `tf.float32` means; make_golden flips it to float64
float32 = 'float32'                 # symbolic: resolved through _dt()
float64 = 'float64'
int32 = _torch.int32
int64 = _torch.int64
bool = _torch.bool                  # noqa: A001


def set_ref_dtype(dt):
    global REF_DTYPE
    REF_DTYPE = dt


def _dt(dtype):
    if dtype is None:
        return None
    if dtype in ('float32', float32):
        return REF_DTYPE
    if dtype == 'float64':
        return _torch.float64
    return dtype


class RefTensor(_torch.Tensor):
    """torch.Tensor with TF-like value semantics for the in-place operators."""

    @staticmethod
    def _c(o):
        return _wrap(o) if isinstance(o, _np.ndarray) else o

    def numpy(self):
        return _torch.Tensor.numpy(self.detach().as_subclass(_torch.Tensor))

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    __array_priority__ = 10000

    def __add__(self, o): return _torch.Tensor.__add__(self, self._c(o))
    def __radd__(self, o): return _torch.Tensor.__radd__(self, self._c(o))
    def __sub__(self, o): return _torch.Tensor.__sub__(self, self._c(o))
    def __rsub__(self, o): return _torch.Tensor.__rsub__(self, self._c(o))
    def __mul__(self, o): return _torch.Tensor.__mul__(self, self._c(o))
    def __rmul__(self, o): return _torch.Tensor.__rmul__(self, self._c(o))
    def __truediv__(self, o): return _torch.Tensor.__truediv__(self, self._c(o))
    def __rtruediv__(self, o): return _torch.Tensor.__rtruediv__(self, self._c(o))
    def __iadd__(self, o): return self + o
    def __isub__(self, o): return self - o
    def __imul__(self, o): return self * o
    def __itruediv__(self, o): return self / o
    def __lt__(self, o): return _torch.Tensor.__lt__(self, self._c(o))
    def __le__(self, o): return _torch.Tensor.__le__(self, self._c(o))
    def __gt__(self, o): return _torch.Tensor.__gt__(self, self._c(o))
    def __ge__(self, o): return _torch.Tensor.__ge__(self, self._c(o))
    __hash__ = _torch.Tensor.__hash__


def _wrap(x, dtype=None):
    dtype = _dt(dtype)
    if isinstance(x, _torch.Tensor):
        t = x if dtype is None or x.dtype == dtype else x.to(dtype)
    elif isinstance(x, (list, tuple)) and len(x) and isinstance(x[0], _torch.Tensor):
        t = _torch.stack([e if isinstance(e, _torch.Tensor) else _torch.as_tensor(e) for e in x])
        if dtype is not None:
            t = t.to(dtype)
    else:
        a = _np.asarray(x)
        if dtype is None:
            if a.dtype == _np.float64 and not isinstance(x, _np.ndarray):
                dtype = REF_DTYPE              # python floats become "float32" like in TF
            elif a.dtype == _np.float32 and REF_DTYPE == _torch.float64:
                dtype = _torch.float64         # fp64 yard-stick mode promotes f32 inputs
        t = _torch.as_tensor(a.copy() if isinstance(x, _np.ndarray) else a)
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
    return t if isinstance(t, RefTensor) else t.as_subclass(RefTensor)


def constant(value, dtype=None, shape=None, name=None):
    return _wrap(value, dtype)


def convert_to_tensor(value, dtype=None, name=None):
    return _wrap(value, dtype)


class Variable(RefTensor):
    @staticmethod
    def __new__(cls, initial_value, dtype=None, trainable=True, name=None):
        t = _wrap(initial_value, dtype).detach().clone().as_subclass(cls)
        t.requires_grad_(True if trainable else False)
        return t

    def __init__(self, *a, **k):
        pass

    def assign(self, value):
        with _torch.no_grad():
            self.copy_(_wrap(value).to(self.dtype))
        return self


def _f(fn):
    def g(x, *a, **k):
        k.pop('name', None)
        return fn(_wrap(x), *a, **k)
    return g


cos = _f(_torch.cos)
sin = _f(_torch.sin)
atan = _f(_torch.atan)
sqrt = _f(_torch.sqrt)
square = _f(_torch.square)
abs = _f(_torch.abs)          # noqa: A001
tanh = _f(_torch.tanh)
exp = _f(_torch.exp)


def pow(x, y, name=None):     # noqa: A001
    x = _wrap(x)
    if not x.is_floating_point():
        x = x.to(REF_DTYPE)
    y = _wrap(y).to(x.dtype) if not isinstance(y, (int, float)) else y
    return _torch.pow(x, y)


def where(cond, x=None, y=None, name=None):
    return _torch.where(_wrap(cond), _wrap(x), _wrap(y))


def zeros(shape, dtype='float32'):
    return _wrap(_torch.zeros(tuple(shape) if not isinstance(shape, int) else (shape,), dtype=_dt(dtype)))


def ones(shape, dtype='float32'):
    return _wrap(_torch.ones(tuple(shape) if not isinstance(shape, int) else (shape,), dtype=_dt(dtype)))


def zeros_like(x, dtype=None):
    return _wrap(_torch.zeros_like(_wrap(x), dtype=_dt(dtype)))


def ones_like(x, dtype=None):
    return _wrap(_torch.ones_like(_wrap(x), dtype=_dt(dtype)))


def stack(values, axis=0, name=None):
    return _torch.stack([_wrap(v) for v in values], dim=axis)


def concat(values, axis=0, name=None):
    return _torch.cat([_wrap(v) for v in values], dim=axis)


def split(value, num_or_size_splits, axis=0):
    value = _wrap(value)
    if isinstance(num_or_size_splits, int):
        return list(_torch.chunk(value, num_or_size_splits, dim=axis))
    return list(_torch.split(value, list(num_or_size_splits), dim=axis))


def tile(x, multiples):
    return _wrap(x).repeat(*[int(m) for m in multiples])


def reshape(x, shape, name=None):
    return _wrap(x).reshape(tuple(int(s) for s in shape))


def squeeze(x, axis=None):
    return _wrap(x).squeeze() if axis is None else _wrap(x).squeeze(axis)


def clip_by_value(x, lo, hi, name=None):
    return _torch.clamp(_wrap(x), lo, hi)


def _red(x):
    return stack(list(x), 0) if isinstance(x, (tuple, list)) else _wrap(x)


def reduce_mean(x, axis=None, keepdims=False):
    x = _red(x)
    return x.mean() if axis is None else x.mean(dim=axis, keepdim=keepdims)


def reduce_sum(x, axis=None, keepdims=False):
    x = _red(x)
    return x.sum() if axis is None else x.sum(dim=axis, keepdim=keepdims)


def reduce_min(x, axis=None):
    x = _red(x)
    return x.min() if axis is None else x.min(dim=axis).values


def stop_gradient(x):
    return _wrap(x).detach()


def shape(x):
    return tuple(_wrap(x).shape)


def matmul(a, b):
    return _torch.matmul(_wrap(a), _wrap(b))


@contextlib.contextmanager
def name_scope(name):
    yield name


def function(fn=None, **kwargs):
    if fn is None:
        return lambda f: f
    return fn


class GradientTape(object):
    def __init__(self, persistent=False, watch_accessed_variables=True):
        self.persistent = persistent

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def watch(self, x):
        pass

    def gradient(self, target, sources):
        single = isinstance(sources, _torch.Tensor)
        srcs = [sources] if single else list(sources)
        gs = _torch.autograd.grad(target, srcs, retain_graph=True, allow_unused=True)
        gs = [_wrap(_torch.zeros_like(s)) if g is None else _wrap(g) for g, s in zip(gs, srcs)]
        return gs[0] if single else gs


def clip_by_global_norm(t_list, clip_norm):
    """TF semantics (SURVEY A-5): n = sqrt(sum ||g||^2); g * clip * min(1/n, 1/clip)."""
    t_list = [_wrap(t) for t in t_list]
    n = _torch.sqrt(sum((t * t).sum() for t in t_list))
    scale = clip_norm * _torch.minimum(1.0 / n, _torch.as_tensor(1.0 / clip_norm, dtype=n.dtype))
    return [t * scale for t in t_list], n


class Module(object):
    def __init__(self, name=None):
        self._name = name


# ---- sub-namespaces --------------------------------------------------------------------------
def _ns(name, **kw):
    m = types.ModuleType(__name__ + '.' + name)
    m.__dict__.update(kw)
    sys.modules[m.__name__] = m
    return m


def _reduce_variance(x, axis=None):
    x = _wrap(x)
    return x.var(unbiased=False) if axis is None else x.var(dim=axis, unbiased=False)


math = _ns('math', reduce_variance=_reduce_variance)
nn = _ns('nn', softmax=lambda x, axis=-1: _torch.softmax(_wrap(x), dim=axis))
linalg = _ns('linalg', inv=lambda x: _torch.linalg.inv(_wrap(x)))

_noise_source = None


def set_noise_source(fn):
    """fn(shape) -> standard-normal numpy array/tensor; None restores torch.randn."""
    global _noise_source
    _noise_source = fn


def _std_normal(shp):
    shp = tuple(int(s) for s in shp)
    if _noise_source is None:
        return _wrap(_torch.randn(shp, dtype=REF_DTYPE))
    return _wrap(_noise_source(shp)).to(REF_DTYPE).reshape(shp)


def _random_normal(shape, mean=0.0, stddev=1.0, dtype='float32', seed=None, name=None):
    return _std_normal(shape) * stddev + mean


random = _ns('random', normal=_random_normal)


class _Noop(object):
    def __getattr__(self, k):
        return _Noop()

    def __call__(self, *a, **k):
        return _Noop()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


config = _Noop()
summary = _Noop()


class _Checkpoint(object):
    """tf.train.Checkpoint: accepted and ignored.  SingleProcessOffPolicyOptimizer.step saves at iteration 0
    (optimizer.py:385-387 -> policy.py:98-103); the wire format is outside the oracle surface, so save / restore do nothing."""

    def __init__(self, **kw):
        self.kw = kw

    def save(self, path):
        return None

    def restore(self, path):
        return None


train = _ns('train', Checkpoint=_Checkpoint)

from . import keras  # noqa: E402,F401
