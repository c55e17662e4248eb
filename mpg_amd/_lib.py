"""ctypes binding of libmpg_hip.so.  There is NO CPU fallback: if the library is missing, or a call fails,
an exception is raised (the product path never routes through oracle/)."""
import ctypes
import os
import re

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libmpg_hip.so')
HEADER = os.path.join(HERE, '..', 'include', 'mpg_hip.h')
# The library is built twice from the same sources (mpg_amd/build.py): the product, whose 256 x 256 hidden-layer products run
# as split-fp16 operands on the f16 matrix pipe (csrc/mlp_core.h), and the EXACT-fp32 engine (-DMPG_F32_MFMA:
# v_mfma_f32_16x16x4_f32, no fp16 operand anywhere) that the split engine is measured and tested against (model.py:39-43 is
# plain float32).  Same ABI, same entry points; the packed weight images differ, so objects built under one engine stay there.
ENGINES = {'split': LIB_PATH, 'f32': os.path.join(HERE, 'libmpg_hip_f32.so')}

_libs = {}
_engine = 'split'
ABI_VERSION = 10      # the struct mirrors of ops.py (CfgStruct, WCacheStruct, ...) follow include/mpg_hip.h at this version


class MpgError(RuntimeError):
    pass


def _header_text():
    src = open(HEADER).read()
    return re.sub(r'/\*.*?\*/', '', src, flags=re.S)


def declared_symbols():
    """Every function name include/mpg_hip.h declares (used by the ABI export test)."""
    return sorted(set(re.findall(r'\b(mpg_[a-z0-9_]+)\s*\(', _header_text())))


def declared_return_types():
    """{function name: C return type as written in include/mpg_hip.h} - the restype of every symbol is taken from
    the declaration itself, not from a naming convention."""
    out = {}
    for m in re.finditer(r'^\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\*?)\s*(mpg_[a-z0-9_]+)\s*\(', _header_text(), flags=re.M):
        out[m.group(2)] = re.sub(r'\s+', ' ', m.group(1)).strip()
    return out


def lib():
    """the library of the engine in force (select_engine / engine(); default: the split-fp16 product)"""
    h = _libs.get(_engine)
    if h is None:
        path = ENGINES[_engine]
        if not os.path.exists(path):
            raise MpgError('%s is missing - run `python -m mpg_amd.build` (hipcc, gfx950). '
                           'mpg_amd has no CPU fallback.' % path)
        h = ctypes.CDLL(path)
        ctype = {'size_t': ctypes.c_size_t, 'int': ctypes.c_int, 'const char*': ctypes.c_char_p, 'const char *': ctypes.c_char_p}
        rtypes = declared_return_types()
        for name in declared_symbols():
            fn = getattr(h, name)               # AttributeError here = header/library out of sync
            fn.restype = ctype[rtypes[name]]    # KeyError here = a return type this binding does not know
        got = h.mpg_abi_version()
        if got != ABI_VERSION:
            raise MpgError('%s has ABI version %d, this binding mirrors version %d - rebuild with `python -m mpg_amd.build`'
                           % (path, got, ABI_VERSION))
        _libs[_engine] = h
    return h


def select_engine(name):
    """'split' (the product) or 'f32' (libmpg_hip_f32.so, the exact-fp32 engine).  Returns the previous selection.  Objects that
    hold packed weight images (ops.WeightCache, PolicyWithQs) belong to the engine they were created under."""
    global _engine
    assert name in ENGINES, name
    old, _engine = _engine, name
    return old


def current_engine():
    return _engine


class engine(object):
    """with L.engine('f32'): ...   - every library call inside goes to that engine's shared object"""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.old = select_engine(self.name)
        return self

    def __exit__(self, *exc):
        select_engine(self.old)
        return False


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return ctypes.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), 'device-resident contiguous tensors only'
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise MpgError('%s failed (%d): %s' % (what, rc, lib().mpg_last_error().decode()))


def call(name, *args):
    check(getattr(lib(), name)(*args), name)


c_int, c_float, c_double, c_u64, c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_uint64, ctypes.c_size_t
