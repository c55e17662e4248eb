#!/usr/bin/env python3
"""Where does a short bench run spend its time?  Same stack and loop as bench.py, with per-step host timestamps
(enqueue time of opt.step()) and one HIP event per step (GPU-side step spacing).

    python3 tools/bench_trace.py --steps 40 --warmup 5 [--prof 16] [--sync-each]
"""
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
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--prof', type=int, default=0, help='mpg_prof_enable(every) after the warm-up, like bench.py r01')
    ap.add_argument('--sync-each', action='store_true', help='synchronise after every step (isolates GPU time per step)')
    a = ap.parse_args()
    import bench
    import mpg_amd._lib as L
    t_imp = time.perf_counter()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    args, worker, learner, rb, opt = bench.build_stack(dev, seed=0)
    torch.cuda.synchronize()
    t_built = time.perf_counter()
    lib = L.lib()
    wu = []
    for _ in range(a.warmup):
        t = time.perf_counter()
        opt.step()
        torch.cuda.synchronize()
        wu.append(1e3 * (time.perf_counter() - t))
    if a.prof:
        from mpg_amd import ops
        prof = ops.Profiler(max_samples=a.steps)
        opt.set_profiler(prof)
        prof.start(a.prof)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    host = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(a.steps):
        t = time.perf_counter()
        opt.step()
        if a.sync_each:
            torch.cuda.synchronize()
        host.append(1e3 * (time.perf_counter() - t))
        ev[i + 1].record()
    t_enq = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps)]
    print(json.dumps({'build_stack_s': t_built - t_imp, 'warmup_ms_each_synced': [round(x, 3) for x in wu],
                      'timed_total_ms': 1e3 * (t1 - t0), 'enqueue_done_ms': 1e3 * (t_enq - t0),
                      'ms_per_step': 1e3 * (t1 - t0) / a.steps,
                      'host_enqueue_ms': [round(x, 3) for x in host], 'gpu_step_ms': [round(x, 3) for x in gpu]}))


if __name__ == '__main__':
    main()
