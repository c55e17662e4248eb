#!/usr/bin/env python3
"""Scan for one-off stalls: run N synchronised steps right after start-up and print every step slower than 2x the median
with its time since the first GPU call.  python3 tools/stall_scan.py [--steps 6000] [--quiesce-gc]

Round-2 finding (profiles/README.md): the 40-100 ms stalls sit at the SAME step indices run after run (2081: ~1 ms,
2773: 36-98 ms) - Python's generational garbage collector walking the ~10^6 objects `import torch` leaves behind - and are
gone with --quiesce-gc."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=6000)
    ap.add_argument('--idle', type=float, default=0.0, help='sleep this long half-way (does the stall re-appear after idling?)')
    ap.add_argument('--quiesce-gc', action='store_true', help='mpg_amd.optimizer.quiesce_gc() + gc.disable() before the loop')
    a = ap.parse_args()
    import bench
    t_first = time.perf_counter()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    t_init = time.perf_counter()
    args, worker, learner, rb, opt = bench.build_stack(dev, seed=0)
    torch.cuda.synchronize()
    t_built = time.perf_counter()
    if a.quiesce_gc:
        import gc
        from mpg_amd.optimizer import quiesce_gc
        quiesce_gc()
        gc.disable()
    ts = []
    for i in range(a.steps):
        if a.idle and i == a.steps // 2:
            time.sleep(a.idle)
        t = time.perf_counter()
        opt.step()
        torch.cuda.synchronize()
        ts.append((t - t_first, 1e3 * (time.perf_counter() - t)))
    d = sorted(x[1] for x in ts)
    med = d[len(d) // 2]
    out = [(i, round(t, 3), round(ms, 3)) for i, (t, ms) in enumerate(ts) if ms > 2 * med]
    print(json.dumps({'gpu_init_s': t_init - t_first, 'build_stack_s': t_built - t_init, 'median_ms': med, 'p99_ms': d[int(len(d) * .99)],
                      'outliers(step, t_since_first_gpu_call_s, ms)': out,
                      'first10': [round(x[1], 3) for x in ts[:10]], 'last_ms': ts[-1][1]}))


if __name__ == '__main__':
    main()
