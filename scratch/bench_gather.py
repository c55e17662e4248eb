import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpg_amd.buffer import ReplayBuffer
from mpg_amd.config import default_args
for cap in (16384, 100000, 500000):
    args = default_args('MPG-v2', replay_batch_size=4096, max_buffer_size=cap, replay_starts=10)
    rb = ReplayBuffer(args, 0)
    n = cap
    for lo in range(0, n, 50000):
        m = min(50000, n - lo)
        rb.add_batch((torch.randn(m, 6, device='cuda'), torch.randn(m, 2, device='cuda'), torch.randn(m, device='cuda'),
                      torch.randn(m, 6, device='cuda'), torch.ones(m, dtype=torch.uint8, device='cuda')))
    for _ in range(5): rb.replay()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(200): rb.replay()
    ev[1].record(); torch.cuda.synchronize()
    print(cap, 'rows: replay (sample+gather via separate idx + gather kernels) %.2f us' % (ev[0].elapsed_time(ev[1]) * 1000 / 200))
