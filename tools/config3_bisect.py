#!/usr/bin/env python3
"""Which input makes the device's config-3 curves differ from tools/config3_oracle_run.py's (target_max at
iteration 250, device -0.30 on 4/4 seeds vs oracle -0.52 .. -0.65)?  tests/test_noise_gpu.py shows that the device loop follows
tests/c3_loop.py (the oracle's loop on the device's Philox draws, from the device's initial weights) to 2e-5 over 120
iterations, so the question can be answered on the CPU with that loop alone, by changing ONE input at a time:

    init = device    PolicyWithQs' initialisation (mpg_amd/policy.py init_mlp_flat: Orthogonal(sqrt 2 / 1), ZERO biases = model.py:23-36)
    init = jitter    tests/golden_inputs.mlp_weights_flat (the parity fixtures' weights: biases ~ N(0, 0.05)) - what
                     tools/config3_oracle_run.py started from
    dtype            float32 / float64 networks

    python3 tools/config3_bisect.py <init> <dtype> <seed> [iterations=500] [every=50]   -> one JSON line per checkpoint
Test infrastructure (imports oracle/ through tests/c3_loop.py); CPU only."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.c3_loop import OracleConfig3Loop          # noqa: E402
from tests.golden_inputs import mlp_weights_flat     # noqa: E402


def device_init(seed):
    from mpg_amd.policy import init_mlp_flat
    gen = torch.Generator().manual_seed(seed)
    return [init_mlp_flat(gen, 5, 1).numpy(), init_mlp_flat(gen, 4, 2).numpy()]      # names order: Q1, policy


def main():
    init, dt, seed = sys.argv[1], sys.argv[2], int(sys.argv[3])
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 500
    every = int(sys.argv[5]) if len(sys.argv) > 5 else 50
    torch.set_num_threads(int(os.environ.get('THREADS', '2')))
    if init.startswith('device'):
        q, p = device_init(seed)
    else:
        rng = np.random.Generator(np.random.PCG64(seed))
        q, p = mlp_weights_flat(rng, 5, 1), mlp_weights_flat(rng, 4, 2)
        if init == 'jitter_zero_bias':                    # the same kernels with the biases zeroed
            for w, din, dout in ((q, 5, 1), (p, 4, 2)):
                o = din * 256
                w[o:o + 256] = 0
                w[o + 256 + 65536:o + 512 + 65536] = 0
                w[-dout:] = 0
    loop = OracleConfig3Loop(q, p, seed=seed, dtype=torch.float64 if dt == 'float64' else torch.float32)
    t0 = time.time()
    for it in range(0, iters + 1, every):
        ret, th = loop.evaluate()
        rec = dict(init=init, dtype=dt, seed=seed, iteration=it, episode_return=round(ret, 3), theta_rms=round(th, 5), wall_s=round(time.time() - t0, 1))
        if it:
            st = loop.stats
            rec.update(target_max=float(np.max(st['targets'])), target_mean=float(np.mean(st['targets'])), value_mean=float(st['value_mean']),
                       q_loss=float(st['q_loss']), q_gradient_norm=float(st['q_gradient_norm']), policy_gradient_norm=float(st['policy_gradient_norm']))
        print(json.dumps(rec), flush=True)
        if it < iters:
            for _ in range(every):
                loop.step()


if __name__ == '__main__':
    main()
