#!/usr/bin/env python3
"""Times the chain-free network kernels at large batches through the C ABI (HIP events, 100 launches each):
    python3 tools/pp_bench.py            (the build decides: ping-pong kernels from MPG_PP_MIN_GROUPS_PER_WG groups per workgroup on)
A/B: MPG_EXTRA_CFLAGS=-DMPG_PP_MIN_GROUPS_PER_WG=1000000 python3 -m mpg_amd.build --split-only   -> lock-step kernels at every size"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpg_amd import ops   # noqa: E402
from mpg_amd import _lib as L   # noqa: E402


def rand_net(rng, din, dout):
    ws = [rng.standard_normal((din, 256)) * 0.4, rng.standard_normal(256) * 0.1, rng.standard_normal((256, 256)) * (1.4 / 16),
          rng.standard_normal(256) * 0.1, rng.standard_normal((256, dout)) * 0.1, rng.standard_normal(dout) * 0.1]
    return np.concatenate([w.ravel() for w in ws]).astype(np.float32)


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


rng = np.random.Generator(np.random.PCG64(0))
for (din, dout, used, act) in ((8, 1, 1, 0), (6, 4, 2, 1)):
    flat = torch.as_tensor(rand_net(rng, din, dout)).cuda()
    wc = ops.WeightCache(flat, [(din, dout)])
    for rows in (16384, 32768, 65536, 131072):
        x = torch.randn(rows, din, device='cuda')
        y = torch.empty(rows, used, device='cuda')
        sc = (L.ctypes.c_float * 16)(*([1.0] * 16))

        def fwd():
            L.call('mpg_mlp_forward', L.ptr(flat), L.c_int(din), L.c_int(dout), L.c_int(used), L.c_int(act), L.c_int(rows), L.ptr(x), sc,
                   L.c_int(din), L.ptr(y), wc.ref, L.stream())
        print('forward  in=%d out=%d rows %6d: %.1f us' % (din, used, rows, timeit(fwd)), flush=True)
