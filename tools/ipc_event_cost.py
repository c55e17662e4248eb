#!/usr/bin/env python3
"""What an INTERPROCESS event costs on the GPU timeline and on the host (round 6: the one-rank one-shot exchange measured 51 us between
its two timing events although it enqueues one small kernel).  Prints, per kind of event: GPU time per record (two timing events around
N back-to-back records, each separated by a tiny kernel so that the records cannot merge) and host time per record call."""
import time
import torch

N = 200
x = torch.zeros(1024, device='cuda')
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def run(make):
    evs = [make() for _ in range(N)]
    for e in evs:
        e.record()
    torch.cuda.synchronize()
    t0.record()
    h0 = time.perf_counter()
    for e in evs:
        x.add_(1.0)
        e.record()
    h1 = time.perf_counter()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / N, (h1 - h0) * 1e6 / N


def base():
    torch.cuda.synchronize()
    t0.record()
    h0 = time.perf_counter()
    for _ in range(N):
        x.add_(1.0)
    h1 = time.perf_counter()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / N, (h1 - h0) * 1e6 / N


for name, fn in (('kernel only', None), ('plain event (no timing)', lambda: torch.cuda.Event(enable_timing=False)),
                 ('timing event', lambda: torch.cuda.Event(enable_timing=True)),
                 ('interprocess event', lambda: torch.cuda.Event(enable_timing=False, interprocess=True))):
    for rep in range(2):
        g, h = base() if fn is None else run(fn)
        print('%-26s GPU %.2f us per (kernel + record), host %.2f us' % (name, g, h))
