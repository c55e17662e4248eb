#!/usr/bin/env python3
"""k_forward (pairs of groups) vs k_forward_stream at large batches: time per launch and bit-identity of the outputs.
python3 archive/proto/stream_forward/fwd_stream_ab.py   (runs itself twice as a child process: MPG_FORWARD_STREAM=0 / 1)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
SHAPES = [(8, 1, 1, 0), (6, 4, 2, 1), (5, 1, 1, 0)]


def child(out):
    import torch
    from mpg_amd import ops
    from mpg_amd.policy import init_mlp_flat
    res = {}
    for rows in (65536, 65536 + 37, 8192):
        for din, dout, used, act in SHAPES:
            gen = torch.Generator().manual_seed(din)
            flat = init_mlp_flat(gen, din, dout).cuda()
            wc = ops.WeightCache(flat, [(din, dout)])
            x = torch.randn(rows, din, generator=gen).cuda()
            y = ops.mlp_forward(flat, din, dout, used, act, x, wcache=wc)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for _ in range(20):
                ops.mlp_forward(flat, din, dout, used, act, x, wcache=wc)
            ev[0].record()
            for _ in range(200):
                ops.mlp_forward(flat, din, dout, used, act, x, wcache=wc)
            ev[1].record()
            torch.cuda.synchronize()
            us = ev[0].elapsed_time(ev[1]) / 200 * 1e3
            res['y_%d_%d' % (rows, din)] = y.cpu().numpy()
            res['t_%d_%d' % (rows, din)] = us
    np.savez(out, **res)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    outs = []
    for mode in ('0', '1'):
        out = '/tmp/fwd_stream_%s.npz' % mode
        subprocess.check_call([sys.executable, __file__, out], env=dict(os.environ, MPG_FORWARD_STREAM=mode))
        outs.append(np.load(out))
    for k in sorted(outs[0].files):
        if k.startswith('t_'):
            same = np.array_equal(outs[0]['y' + k[1:]], outs[1]['y' + k[1:]])
            print('%-14s pairs %.1f us   stream %.1f us   outputs bit-identical: %s' % (k[2:], outs[0][k], outs[1][k], same))
