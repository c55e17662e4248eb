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
ABI_VERSION = 7      # the struct mirrors of ops.py (CfgStruct, WCacheStruct, ...) follow include/mpg_hip.h at this version


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
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MpgError('%s is missing - run `python -m mpg_amd.build` (hipcc, gfx950). '
                           'mpg_amd has no CPU fallback.' % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        ctype = {'size_t': ctypes.c_size_t, 'int': ctypes.c_int, 'const char*': ctypes.c_char_p, 'const char *': ctypes.c_char_p}
        rtypes = declared_return_types()
        for name in declared_symbols():
            fn = getattr(_lib, name)            # AttributeError here = header/library out of sync
            fn.restype = ctype[rtypes[name]]    # KeyError here = a return type this binding does not know
        got = _lib.mpg_abi_version()
        if got != ABI_VERSION:
            _lib = None
            raise MpgError('%s has ABI version %d, this binding mirrors version %d - rebuild with `python -m mpg_amd.build`'
                           % (LIB_PATH, got, ABI_VERSION))
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
