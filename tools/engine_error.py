#!/usr/bin/env python3
"""Accuracy of the hidden-layer engine against the float64 oracle on the golden nets: forward values, 25-step mixed policy
gradient (BPTT), critic loss gradient.  Run it once per engine build (default split-fp16, -DMPG_F32_MFMA exact fp32)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mpg_oracle as O   # noqa: E402  (tool, not product)


def rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    from mpg_amd import ops
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'mpg_v2_H256_B64.npz')))
    dev = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).cuda()
    cfg, ocfg = ops.make_cfg(), O.Cfg()
    wp, wq = g['w_policy'], g['w_Q1']
    nets = O.Nets(ocfg, {'policy': wp, 'Q1': wq}, dtype=torch.float64)
    obs, act, eps = g['batch_obs'], g['batch_actions'], g['eps']
    B = obs.shape[0]
    # forward
    a = ops.policy_action(cfg, dev(wp), dev(obs)).cpu().numpy()
    ref = nets.compute_action(O.process_obses(ocfg, torch.as_tensor(obs).double())).detach().numpy()
    print('policy forward  rel-L2 vs f64: %.3e' % rel(a, ref))
    x = torch.cat([dev(obs), dev(act)], 1).contiguous()
    q = ops.mlp_forward(dev(wq), 8, 1, 1, 0, x, in_scale=[cfg.obs_scale[i] for i in range(6)], n_scaled=6).cpu().numpy()[:, 0]
    qref = nets.q('Q1', O.process_obses(ocfg, torch.as_tensor(obs).double()), torch.as_tensor(act).double()).detach().numpy()
    print('critic forward  rel-L2 vs f64: %.3e' % rel(q, qref))
    # BPTT gradient
    for w in ([0.4, 0.6], [1e-5, 0.99999]):
        ret, _, grad = ops.rollout_pg(cfg, dev(wp), dev(wq), dev(obs), dev(eps), [0, 25], np.array(w, np.float32))
        reduced, _, _ = O.model_rollout_for_policy_update(ocfg, nets, torch.as_tensor(obs).double(), torch.as_tensor(eps).double())
        loss = -(w[0] * reduced[0] + w[1] * reduced[25])
        gref = np.concatenate([t.numpy().ravel() for t in torch.autograd.grad(loss, nets.w['policy'])])
        got = grad.cpu().numpy()
        o = 0
        parts = []
        for shp in O.mlp_shapes(6, 256, 4):
            n = int(np.prod(shp))
            if np.linalg.norm(gref[o:o + n]) > 0:
                parts.append('%s %.2e' % (shp, rel(got[o:o + n], gref[o:o + n])))
            o += n
        print('mixed PG w=%s: total %.3e | returns %.2e | %s' % (w, rel(got, gref), rel(ret.cpu().numpy() / B, reduced[[0, 25]].detach().numpy()), ' '.join(parts)))
    # critic gradient
    y = np.random.Generator(np.random.PCG64(0)).standard_normal(B).astype(np.float32)
    _, gq, _ = ops.q_loss_grad(cfg, dev(wq), dev(obs), dev(act), dev(y))
    ws = nets.w['Q1']
    qq = O.mlp(ws, torch.cat([O.process_obses(ocfg, torch.as_tensor(obs).double()), torch.as_tensor(act).double()], 1), 'linear')[:, 0]
    l = 0.5 * torch.mean((qq - torch.as_tensor(y).double()) ** 2)
    gref = np.concatenate([t.numpy().ravel() for t in torch.autograd.grad(l, ws)])
    print('critic gradient rel-L2 vs f64: %.3e' % rel(gq.cpu().numpy(), gref))


if __name__ == '__main__':
    main()
