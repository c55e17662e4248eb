import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import mpg_oracle as O
from tests import yardstick as Y
from mpg_amd.buffer import ReplayBuffer
from mpg_amd.config import default_args
from mpg_amd.learners import MPGLearner
from mpg_amd.optimizer import SingleProcessOffPolicyOptimizer
from mpg_amd.policy import PolicyWithQs
from mpg_amd.worker import OffPolicyWorker
DEV = 'cuda'
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
def rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for seed in range(NS):
    torch.manual_seed(seed)
    args = default_args('MPG-v2', num_agent=64, batch_size=512, replay_batch_size=256, replay_starts=1024, max_buffer_size=8192,
                        value_lr_schedule=[1e-3, 100000, 1e-4], num_future_data=K, seed=seed)
    worker = OffPolicyWorker(PolicyWithQs, args.env_id, args, 0)
    learner = MPGLearner(PolicyWithQs, args)
    rb = ReplayBuffer(args, 0)
    opt = SingleProcessOffPolicyOptimizer(worker, learner, rb, None, args, fused=True)
    pw = worker.policy_with_value
    for i in range(30):
        opt.step()
    batch = [b.clone() for b in rb.sample(256)[:5]]
    eps = torch.randn(25, 256, device=DEV)
    learner.counter = 0
    grads = learner.compute_gradient(batch, None, None, 500, eps=eps)
    got = torch.cat([x.reshape(-1) for x in grads]).cpu().numpy()
    od = 6 + K
    cfg = O.Cfg(obs_dim=od, obs_scale=list(O.OBS_SCALE_PT) + [1.] * K)
    flat, tflat = pw.params.cpu().numpy(), pw.targets.cpu().numpy()
    off = np.cumsum([0] + list(pw.sizes))
    w = {n: flat[off[i]:off[i + 1]] for i, n in enumerate(pw.names)}
    wt = {n: tflat[off[i]:off[i + 1]] for i, n in enumerate(pw.names)}
    nb = [b.cpu().numpy() for b in batch]
    ref = {}
    for dt in (torch.float32, torch.float64):
        nets = O.Nets(cfg, w, flat_targets=wt, dtype=dt)
        g, _ = O.mpg_compute_gradient(cfg, nets, nb, eps.cpu().numpy(), 500, 'MPG-v2')
        ref[dt] = [np.asarray(x, np.float64) for x in g]
    o = 0
    line = []
    for i, (a32, a64) in enumerate(zip(ref[torch.float32], ref[torch.float64])):
        n = a64.size
        e_got, e_ref = rel(got[o:o + n], a64), rel(a32, a64)
        if a64.size >= 8:
            line.append('%s%.1f(%.0e)' % ('!' if e_got > 4 * e_ref + 1e-7 else '', e_got / e_ref, e_got))
        o += n
    print('seed', seed, 'K', K, ' | '.join(line))
