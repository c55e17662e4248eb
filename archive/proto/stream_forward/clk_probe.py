import torch, sys
sys.path.insert(0, '.')
from mpg_amd import ops
from mpg_amd.policy import init_mlp_flat
gen = torch.Generator().manual_seed(1)
flat = init_mlp_flat(gen, 8, 1).cuda()
wc = ops.WeightCache(flat, [(8, 1)])
x = torch.randn(65536, 8, generator=gen).cuda()
for _ in range(30):
    y = ops.mlp_forward(flat, 8, 1, 1, 0, x, wcache=wc)
torch.cuda.synchronize()
