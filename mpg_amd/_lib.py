"""ctypes binding of libmpg_hip.so.  There is NO CPU fallback: if the library is missing, or a call fails,
an exception is raised (the product path never routes through oracle/)."""
import ctypes
import os
import re

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libmpg_hip.so')
HEADER = os.path.join(HERE, '..', 'include', 'mpg_hip.h')

_lib = None


class MpgError(RuntimeError):
    pass


def declared_symbols():
    """Every function name include/mpg_hip.h declares (used by the ABI export test)."""
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mpg_[a-z0-9_]+)\s*\(', src)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MpgError('%s is missing - run `python -m mpg_amd.build` (hipcc, gfx950). '
                           'mpg_amd has no CPU fallback.' % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.mpg_last_error.restype = ctypes.c_char_p
        for name in declared_symbols():
            fn = getattr(_lib, name)            # AttributeError here = header/library out of sync
            if name.endswith('_workspace_bytes'):
                fn.restype = ctypes.c_size_t
            elif name == 'mpg_prof_slot_name':
                fn.restype = ctypes.c_char_p
            elif name != 'mpg_last_error':
                fn.restype = ctypes.c_int
    return _lib


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
