"""PolicyWithQs - device-resident mirror of policy.py:19-245 (deterministic policy branch).

All networks live in ONE flat float32 tensor in the order of the reference's `self.models`
(policy.py:72-86): Q1, (Q2,) policy; the target nets in a second flat tensor in the same order; Adam moments
likewise.  Per-optimizer step counters and the PolynomialDecay schedules (policy.py:54-70) stay on the host.
"""
import math

import ctypes

import numpy as np
import torch

from . import _lib as L
from . import ops


def _orthogonal(gen, rows, cols, gain):
    a = torch.randn(max(rows, cols), min(rows, cols), generator=gen, dtype=torch.float64)
    q, r = torch.linalg.qr(a)
    q = q * torch.sign(torch.diagonal(r))
    if rows < cols:
        q = q.T
    return (gain * q[:rows, :cols]).to(torch.float32)


def init_mlp_flat(gen, in_dim, out_dim, hidden=256):
    """MLPNet initialisation, model.py:23-36: Orthogonal(sqrt 2) hidden kernels, Orthogonal(1) output, zero biases."""
    parts = [_orthogonal(gen, in_dim, hidden, math.sqrt(2.)), torch.zeros(hidden),
             _orthogonal(gen, hidden, hidden, math.sqrt(2.)), torch.zeros(hidden),
             _orthogonal(gen, hidden, out_dim, 1.), torch.zeros(out_dim)]
    return torch.cat([p.reshape(-1) for p in parts])


def mlp_shapes(in_dim, out_dim, hidden=256):
    return [(in_dim, hidden), (hidden,), (hidden, hidden), (hidden,), (hidden, out_dim), (out_dim,)]


def polynomial_decay(sched, step):
    """tf.keras.optimizers.schedules.PolynomialDecay(lr0, decay_steps, lr_end) (policy.py:54,62), evaluated like TensorFlow does:
    in float32, p = min(step, S) / S; lr = (lr0 - lr_end) * (1 - p) + lr_end."""
    f = np.float32
    lr0, S, lr_end = f(sched[0]), f(sched[1]), f(sched[2])
    p = min(f(step), S) / S
    return (lr0 - lr_end) * (f(1) - p) + lr_end


def adam_step_size(sched, steps_done):
    """ApplyAdam's alpha = lr * sqrt(1 - beta_2^t) / (1 - beta_1^t), t = steps_done + 1, with every operand a float32 like in Keras
    (`_prepare_local`: the betas are float32 hyper-parameters, beta^t a float32 pow).  float32(0.999) > 0.999 puts this 6.7e-6 below the
    real-number formula during the first thousands of steps.  The same code as train_step.cpp:adam_step_size."""
    f = np.float32
    lr = polynomial_decay(sched, steps_done)
    t = steps_done + 1
    b1p, b2p = f(float(f(0.9)) ** t), f(float(f(0.999)) ** t)      # correctly rounded float32 powers
    return float(f(lr * np.sqrt(f(1) - b2p) / (f(1) - b1p)))


class PolicyWithQs(object):
    def __init__(self, obs_dim, act_dim, value_lr_schedule=(8e-5, 100000, 8e-6), policy_lr_schedule=(3e-5, 100000, 3e-6),
                 double_Q=True, target=True, tau=0.005, delay_update=2, deterministic_policy=True, action_range=None,
                 policy_out_activation='tanh', env_id='PathTracking-v0', obs_scale=None, rew_scale=None, rew_shift=0.,
                 gamma=0.98, value_num_hidden_units=256, policy_num_hidden_units=256, policy_only=False,
                 device='cuda', seed=0, init_seed=0, **kwargs):
        assert value_num_hidden_units == 256 and policy_num_hidden_units == 256, 'kernels are built for 2x256 nets'
        assert deterministic_policy and not policy_only and target, 'hot-path scope: deterministic actor-critic with targets'
        self.device = torch.device(device)
        self.double_Q, self.tau, self.delay_update = bool(double_Q), float(tau), int(delay_update)
        self.cfg = ops.make_cfg(env_id, obs_scale=obs_scale, rew_scale=rew_scale, rew_shift=rew_shift, gamma=gamma,
                                policy_out_activation=policy_out_activation, action_range=action_range, obs_dim=obs_dim)
        self.obs_dim, self.act_dim = obs_dim, act_dim
        self.names = ['Q1', 'Q2', 'policy'] if self.double_Q else ['Q1', 'policy']      # policy.py:72-86
        self.dims = {'Q1': (obs_dim + act_dim, 1), 'Q2': (obs_dim + act_dim, 1), 'policy': (obs_dim, 2 * act_dim)}
        self.sizes = [ops.net_size(*self.dims[n]) for n in self.names]
        self.offsets = np.cumsum([0] + self.sizes)
        # `init_seed` (not the per-process `seed` that drives exploration / replay / model noise) seeds the initial
        # weights, so that data-parallel replicas start identical
        gen = torch.Generator().manual_seed(init_seed)
        flat = torch.cat([init_mlp_flat(gen, *self.dims[n]) for n in self.names])
        self.params = flat.to(self.device)
        self.targets = self.params.clone()                                             # policy.py:60,68
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.schedules = {n: (tuple(policy_lr_schedule) if n == 'policy' else tuple(value_lr_schedule)) for n in self.names}
        self.opt_steps = {n: 0 for n in self.names}
        self.nonfinite = torch.zeros(len(self.names), dtype=torch.int32, device=self.device)
        # sticky MPG_STATUS_* word (include/mpg_hip.h, "Numerical envelope"): every forward pass made with self.cfg and every
        # (re)pack of the weight images ORs into it; check_status() reads, clears and raises
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.cfg.status = self.status.data_ptr()
        self._bind_weight_cache()

    # ---- weight cache (packed register images of the hidden kernels; caller-owned, see include/mpg_hip.h) ----
    def _bind_weight_cache(self):
        dims = [self.dims[n] for n in self.names]
        self.wc_params = ops.WeightCache(self.params, dims, status=self.status)
        self.wc_targets = ops.WeightCache(self.targets, dims, status=self.status)
        self.cfg.wcache[0] = self.wc_params.pointer
        self.cfg.wcache[1] = self.wc_targets.pointer

    def check_status(self, clear=True):
        """Host read of the status word (synchronises).  Raises MpgError if the split-fp16 engine left its envelope since
        the last check: the results computed in between are not to be trusted (the analogue of the reference's
        `judge_is_nan` stop, optimizer.py:357-361, for a failure TensorFlow's float32 graph cannot have)."""
        bits = int(self.status.item())
        if clear and bits:
            self.status.zero_()
        if bits:
            what = []
            if bits & ops.STATUS_ACTIVATION_RANGE:
                what.append('a first-hidden-layer activation reached 4094 (fp16 image overflow: that row is invalid)')
            if bits & ops.STATUS_PARAMETER_RANGE:
                what.append('a network parameter reached 1023.5 (clamped in the packed image)')
            if bits & ops.STATUS_NAN:
                what.append('NaN in an observation or action (worker.py:95-107 judge_is_nan)')
            if bits & ~ops.STATUS_NAN:
                raise L.MpgError('hidden-layer engine left its numerical envelope: ' + '; '.join(what) +
                                 ' - rebuild with MPG_EXTRA_CFLAGS=-DMPG_F32_MFMA for the exact-fp32 engine')
            raise L.MpgError('; '.join(what))
        return 0

    def refresh_weight_cache(self):
        """call after writing params/targets by anything other than apply_gradients"""
        self.wc_params.pack()
        self.wc_targets.pack()

    def sync_from_rank0(self):
        """Data-parallel start-up: every replica takes rank 0's parameters / targets / optimizer state."""
        import torch.distributed as dist
        if dist.is_initialized() and dist.get_world_size() > 1:
            for t in (self.params, self.targets, self.m, self.v):
                dist.broadcast(t, src=0)
            self.refresh_weight_cache()

    # ---- views ----
    def net(self, name, target=False):
        i = self.names.index(name)
        src = self.targets if target else self.params
        return src[self.offsets[i]:self.offsets[i + 1]]

    def _as_list(self, flat, name):
        out, o = [], 0
        for shp in mlp_shapes(*self.dims[name]):
            n = int(np.prod(shp))
            out.append(flat[o:o + n].view(shp))
            o += n
        return out

    def get_weights(self):
        """[models..., target_models...] each a list of 6 arrays (policy.py:112-114).  COPIES, like Keras' get_weights():
        writing into them does not touch the live parameters (whose packed images would otherwise go stale)."""
        return [self._as_list(self.net(n).clone(), n) for n in self.names] + \
               [self._as_list(self.net(n, True).clone(), n) for n in self.names]

    def set_weights(self, weights):
        """policy.py:116-121"""
        k = len(self.names)
        for i, w in enumerate(weights):
            name = self.names[i % k]
            dst = self.net(name, target=i >= k)
            flat = torch.cat([torch.as_tensor(np.asarray(a.cpu() if isinstance(a, torch.Tensor) else a),
                                              dtype=torch.float32).reshape(-1) for a in w])
            dst.copy_(flat.to(self.device))
        self.refresh_weight_cache()

    def set_flat(self, params, targets=None):
        self.params.copy_(torch.as_tensor(params, dtype=torch.float32).to(self.device))
        if targets is not None:
            self.targets.copy_(torch.as_tensor(targets, dtype=torch.float32).to(self.device))
        self.refresh_weight_cache()

    # ---- forward helpers ----
    def compute_action(self, obs):
        """policy.py:193-204: returns (action, logp=0.).  obs is the PROCESSED obs in the reference; here the 'scale'
        preprocessing is fused into the kernel, so pass RAW obs."""
        return ops.policy_action(self.cfg, self.net('policy'), obs), 0.

    def compute_target_action(self, obs):
        return ops.policy_action(self.cfg, self.net('policy', True), obs), 0.

    def _q(self, name, target, obs, act):
        x = torch.cat([obs, act], 1).contiguous()
        sc = [self.cfg.obs_scale[i] for i in range(self.obs_dim)]
        return ops.mlp_forward(self.net(name, target), self.obs_dim + self.act_dim, 1, 1, ops.ACT_LINEAR, x, in_scale=sc,
                               n_scaled=self.obs_dim, wcache=self.wc_targets if target else self.wc_params)[:, 0]

    def compute_Q1(self, obs, act):
        return self._q('Q1', False, obs, act)

    def compute_Q2(self, obs, act):
        return self._q('Q2', False, obs, act)

    def compute_Q1_target(self, obs, act):
        return self._q('Q1', True, obs, act)

    def compute_Q2_target(self, obs, act):
        return self._q('Q2', True, obs, act)

    # ---- optimizer ----
    def apply_gradients(self, iteration, grads):
        """policy.py:123-156: Adam on the critics every call; policy Adam + Polyak on all targets only when
        iteration % delay_update == 0.  grads: flat tensor (or list of arrays) in the order Q1,(Q2),policy."""
        if not isinstance(grads, torch.Tensor):
            grads = torch.cat([torch.as_tensor(np.asarray(g.cpu() if isinstance(g, torch.Tensor) else g),
                                               dtype=torch.float32).reshape(-1) for g in grads]).to(self.device)
        delayed = int(iteration) % self.delay_update == 0
        lr_t, do_adam, do_polyak = [], [], []
        for n in self.names:
            upd = (n != 'policy') or delayed
            t = self.opt_steps[n] + 1
            lr_t.append(adam_step_size(self.schedules[n], self.opt_steps[n]))
            do_adam.append(int(upd))
            do_polyak.append(int(delayed))
            if upd:
                self.opt_steps[n] = t
        ops.adam_polyak(self.params, self.m, self.v, self.targets, grads, self.sizes, lr_t, do_adam, do_polyak, self.tau,
                        skip_flag=self.nonfinite, wc_w=self.wc_params, wc_target=self.wc_targets)

    # ---- checkpoint (flat blob) ----
    def state_dict(self):
        return dict(params=self.params.cpu(), targets=self.targets.cpu(), m=self.m.cpu(), v=self.v.cpu(),
                    opt_steps=dict(self.opt_steps), names=list(self.names))

    def load_state_dict(self, sd):
        assert sd['names'] == self.names
        for k in ('params', 'targets', 'm', 'v'):
            getattr(self, k).copy_(sd[k].to(self.device))
        self.opt_steps = dict(sd['opt_steps'])
        self.refresh_weight_cache()

    def save_weights(self, save_dir, iteration):
        torch.save(self.state_dict(), '%s/ckpt_ite%d.pt' % (save_dir, iteration))

    def load_weights(self, load_dir, iteration):
        self.load_state_dict(torch.load('%s/ckpt_ite%d.pt' % (load_dir, iteration)))
