import sys, numpy as np, torch
sys.path.insert(0, '.')
import mpg_amd._lib as L
DEV='cuda'
cap, n = 1024, 700
s = torch.empty(2 * cap, dtype=torch.float64, device=DEV)
m = torch.empty(2 * cap, dtype=torch.float64, device=DEV)
stamp = torch.empty(cap, dtype=torch.int32, device=DEV)
L.call('mpg_per_init', L.ptr(s), L.ptr(m), L.ptr(stamp), L.c_int(cap), L.stream())
torch.cuda.synchronize(); print('init ok', s[:4], stamp[:4], flush=True)
idx = torch.arange(n, dtype=torch.int32, device=DEV)
prio = torch.rand(n, device=DEV) + 0.1
L.call('mpg_per_update', L.ptr(s), L.ptr(m), L.ptr(stamp), L.c_int(cap), L.c_int(n), L.ptr(idx), L.ptr(prio), L.c_float(0.6), L.c_float(0.0), L.ptr(None), L.stream())
torch.cuda.synchronize(); print('update ok', s[1].item(), flush=True)
