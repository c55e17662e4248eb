"""Thin typed wrappers over the C ABI (include/mpg_hip.h).  Tensors in, tensors out; no arithmetic here."""
import ctypes

import torch

from . import _lib as L

ACT_LINEAR, ACT_TANH = 0, 1
STATUS_ACTIVATION_RANGE, STATUS_PARAMETER_RANGE, STATUS_NAN = 1, 2, 4      # MPG_STATUS_* (include/mpg_hip.h, "Numerical envelope")
HIDDEN = 256


class WCacheStruct(ctypes.Structure):
    """mpg_wcache_t: caller-owned descriptor of the packed W2 images of one flat parameter vector"""
    _fields_ = [('params', ctypes.c_void_p), ('packed', ctypes.c_void_p), ('n_nets', ctypes.c_int),
                ('in_dim', ctypes.c_int * 8), ('out_dim', ctypes.c_int * 8), ('status', ctypes.c_void_p)]


class CfgStruct(ctypes.Structure):
    """mpg_cfg_t"""
    _fields_ = [('obs_dim', ctypes.c_int), ('act_dim', ctypes.c_int), ('policy_out_act', ctypes.c_int),
                ('action_range', ctypes.c_float), ('obs_scale', ctypes.c_float * 16),
                ('rew_scale', ctypes.c_float), ('rew_shift', ctypes.c_float), ('gamma', ctypes.c_float),
                ('env_kind', ctypes.c_int),
                ('wcache', ctypes.POINTER(WCacheStruct) * 2), ('prof', ctypes.c_void_p), ('status', ctypes.c_void_p),
                ('grad_opts', ctypes.c_void_p)]      # mpg_grad_opts_t*: set by the native step driver only (NULL here)


class WeightCache(object):
    """Owns the packed images of one flat [net0 | net1 | ...] tensor (mpg_wcache_t + the device array).  Keep the object
    alive for as long as a cfg points at it; call pack() after writing `params` by anything but the Adam entry points."""

    def __init__(self, params, dims, status=None):
        k = len(dims)
        self.params = params
        self.status = status           # int32[1] device tensor the (re)packing reports MPG_STATUS_* bits into (keeps it alive)
        self.packed = torch.empty(L.lib().mpg_weight_cache_floats(L.c_int(k)), dtype=torch.float32, device=params.device)
        self.desc = WCacheStruct()
        self.desc.params, self.desc.packed, self.desc.n_nets = params.data_ptr(), self.packed.data_ptr(), k
        self.desc.status = status.data_ptr() if status is not None else None
        for i, (ind, outd) in enumerate(dims):
            self.desc.in_dim[i], self.desc.out_dim[i] = ind, outd
        self.pack()

    def pack(self):
        L.call('mpg_weight_cache_pack', ctypes.byref(self.desc), L.stream())

    @property
    def ref(self):
        return ctypes.byref(self.desc)

    @property
    def pointer(self):
        return ctypes.pointer(self.desc)


def _wc(cache):
    return cache.ref if cache is not None else None


class Profiler(object):
    """mpg_prof_t: caller-owned kernel timer; attach() makes the calls issued with that cfg report to it."""

    def __init__(self, max_samples=4096):
        self.h = ctypes.c_void_p(0)
        self._attached = []          # every cfg that carries this handle: cleared in close(), so none is left dangling
        L.call('mpg_prof_create', L.c_int(max_samples), ctypes.byref(self.h))

    def attach(self, *cfgs):
        for c in cfgs:
            c.prof = self.h.value
            if not any(c is a for a in self._attached):
                self._attached.append(c)

    def detach(self, *cfgs):
        for c in cfgs:
            if c.prof == self.h.value:
                c.prof = None
            self._attached = [a for a in self._attached if a is not c]

    def start(self, every):
        L.call('mpg_prof_start', self.h, L.c_int(every))

    def stop(self):
        L.call('mpg_prof_start', self.h, L.c_int(0))

    def read(self, slot):
        """(average ms per timed launch or None, number of timed launches)"""
        ms, cnt = ctypes.c_double(0), ctypes.c_int(0)
        L.call('mpg_prof_read', self.h, L.c_int(slot), ctypes.byref(ms), ctypes.byref(cnt))
        return (ms.value / cnt.value if cnt.value else None), cnt.value

    def close(self):
        if self.h:
            for c in self._attached:      # a cfg must never outlive the events it points at
                if c.prof == self.h.value:
                    c.prof = None
            self._attached = []
            L.lib().mpg_prof_destroy(self.h)
            self.h = ctypes.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_cfg(env_id='PathTracking-v0', obs_scale=None, rew_scale=None, rew_shift=0.0, gamma=0.98,
             policy_out_activation=None, action_range=None, obs_dim=None):
    """Defaults are the reference's (train_script.py:202-306 / train_script4mujoco.py:296-411).  obs_dim: 6 + num_future_data
    for PathTracking (train_script.py:794-811), up to 16 (num_future_data <= 10, the env's own limit here)."""
    pt = env_id == 'PathTracking-v0'
    c = CfgStruct()
    c.obs_dim, c.act_dim = (int(obs_dim) if (pt and obs_dim) else 6, 2) if pt else (4, 1)
    if pt and not 6 <= c.obs_dim <= 16:
        # the library's own answer for such a cfg is MPG_EINVAL at the first launch (cfg_ok): raise it where the cfg is built
        raise L.MpgError('MPG_EINVAL: PathTracking observations have 6 + num_future_data entries and the env and network kernels serve '
                         'num_future_data <= 10 (policy inputs up to 16 wide, critic inputs up to 18; got obs_dim %d)' % c.obs_dim)
    if policy_out_activation is None:
        policy_out_activation = 'tanh' if pt else 'linear'
    c.policy_out_act = ACT_TANH if policy_out_activation == 'tanh' else ACT_LINEAR
    if action_range is None:
        action_range = 0.0 if pt else 3.0
    c.action_range = float(action_range or 0.0)
    sc = obs_scale if obs_scale is not None else ([1., 1., 2., 1., 2.4, 1 / 1200] if pt else [0.001, 1 / 3, 0.1, 0.5])
    for i in range(16):
        c.obs_scale[i] = float(sc[i]) if i < len(sc) else 1.0
    c.rew_scale = float(rew_scale if rew_scale is not None else (0.01 if pt else 1.0))
    c.rew_shift = float(rew_shift)
    c.gamma = float(gamma)
    c.env_kind = 0 if pt else 1
    return c


def net_size(in_dim, out_dim):
    return in_dim * HIDDEN + HIDDEN + HIDDEN * HIDDEN + HIDDEN + HIDDEN * out_dim + out_dim


def policy_size(cfg):
    return net_size(cfg.obs_dim, 2 * cfg.act_dim)


def q_size(cfg):
    return net_size(cfg.obs_dim + cfg.act_dim, 1)


class Workspace(object):
    """Caller-owned scratch the ABI asks for (grown on demand, reused across calls on one stream)."""

    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, nbytes):
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=self.device)
        return self.buf


_WS = {}


def workspace(device, nbytes, slot=0):
    key = (str(device), slot)
    if key not in _WS:
        _WS[key] = Workspace(device)
    return _WS[key].get(nbytes)


def _f32(t):
    assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous(), (t.dtype, t.device, t.is_contiguous())
    return t


def mlp_forward(params, in_dim, out_dim, out_used, out_act, x, in_scale=None, n_scaled=0, wcache=None):
    rows = x.shape[0]
    y = torch.empty(rows, out_used, dtype=torch.float32, device=x.device)
    sc = (ctypes.c_float * 16)(*([float(v) for v in in_scale] + [1.0] * (16 - len(in_scale)))) if in_scale is not None else None
    L.call('mpg_mlp_forward', L.ptr(_f32(params)), L.c_int(in_dim), L.c_int(out_dim), L.c_int(out_used),
           L.c_int(out_act), L.c_int(rows), L.ptr(_f32(x)), sc, L.c_int(n_scaled), L.ptr(y), _wc(wcache), L.stream())
    return y


def policy_action(cfg, policy_params, obs, explore_sigma=0.0, seed=0, ctr=0):
    rows = obs.shape[0]
    act = torch.empty(rows, cfg.act_dim, dtype=torch.float32, device=obs.device)
    L.call('mpg_policy_action', ctypes.byref(cfg), L.ptr(_f32(policy_params)), L.c_int(rows), L.ptr(_f32(obs)),
           L.c_float(explore_sigma), L.c_u64(seed), L.c_u64(ctr), L.ptr(act), L.stream())
    return act


def q_targets(cfg, policy_t, q1t, q2t, rew, obs_tp1, smooth_eps=None, smooth_sigma=0.2, smooth_clip=0.5):
    rows = obs_tp1.shape[0]
    y = torch.empty(rows, dtype=torch.float32, device=obs_tp1.device)
    nb = L.lib().mpg_q_targets_workspace_bytes(ctypes.byref(cfg), L.c_int(rows))
    ws = workspace(obs_tp1.device, nb)
    L.call('mpg_q_targets', ctypes.byref(cfg), L.ptr(_f32(policy_t)), L.ptr(_f32(q1t)),
           L.ptr(_f32(q2t) if q2t is not None else None), L.c_int(rows), L.ptr(_f32(rew)), L.ptr(_f32(obs_tp1)),
           L.ptr(_f32(smooth_eps) if smooth_eps is not None else None), L.c_float(smooth_sigma),
           L.c_float(smooth_clip), L.ptr(y), L.ptr(ws), L.c_size_t(ws.numel()), L.stream())
    return y


def td3_targets(cfg, policy_t, q1t, q2t, rew, obs_tp1, smooth_eps, smooth_sigma=0.2, smooth_clip=0.5):
    """(y, y1): the smoothed clipped double-Q target and the plain Q1 target of the priorities' td error from ONE evaluation of
    the target policy (mpg_td3_targets; td3.py:69-92)."""
    rows = obs_tp1.shape[0]
    y = torch.empty(rows, dtype=torch.float32, device=obs_tp1.device)
    y1 = torch.empty(rows, dtype=torch.float32, device=obs_tp1.device)
    nb = L.lib().mpg_q_targets_workspace_bytes(ctypes.byref(cfg), L.c_int(rows))
    ws = workspace(obs_tp1.device, nb)
    L.call('mpg_td3_targets', ctypes.byref(cfg), L.ptr(_f32(policy_t)), L.ptr(_f32(q1t)), L.ptr(_f32(q2t)), L.c_int(rows),
           L.ptr(_f32(rew)), L.ptr(_f32(obs_tp1)), L.ptr(_f32(smooth_eps) if smooth_eps is not None else None),
           L.c_float(smooth_sigma), L.c_float(smooth_clip), L.ptr(y), L.ptr(y1), L.ptr(ws), L.c_size_t(ws.numel()), L.stream())
    return y, y1


def normal_fill(n, seed, ctr, device):
    """n standard normals from the library's Philox stream (mpg_normal_fill)"""
    out = torch.empty(n, dtype=torch.float32, device=device)
    L.call('mpg_normal_fill', L.c_int(n), L.c_u64(seed), L.c_u64(ctr), L.ptr(out), L.stream())
    return out


def td3_priority_errors(y1, y, td):
    """(y1 - y) - td  = y1 - Q1(s, a), the priorities' td error of TD3 (td3.py:83-92) from the critic pass's own td output"""
    out = torch.empty_like(td)
    L.call('mpg_td3_priority_errors', L.c_int(td.numel()), L.ptr(_f32(y1)), L.ptr(_f32(y)), L.ptr(_f32(td)), L.ptr(out), L.stream())
    return out


def nstep_targets(cfg, policy_t, q1t, rewards, last_obs):
    n, rows = rewards.shape
    y = torch.empty(rows, dtype=torch.float32, device=last_obs.device)
    nb = L.lib().mpg_q_targets_workspace_bytes(ctypes.byref(cfg), L.c_int(rows))
    ws = workspace(last_obs.device, nb)
    L.call('mpg_nstep_targets', ctypes.byref(cfg), L.ptr(_f32(policy_t)), L.ptr(_f32(q1t)), L.c_int(rows), L.c_int(n),
           L.ptr(_f32(rewards)), L.ptr(_f32(last_obs)), L.ptr(y), L.ptr(ws), L.c_size_t(ws.numel()), L.stream())
    return y


def q_loss_grad(cfg, q_params, obs, act, y, inv_b_global=None, grad_out=None, loss_out=None, want_td=False):
    rows = obs.shape[0]
    dev = obs.device
    grad = grad_out if grad_out is not None else torch.empty(q_size(cfg), dtype=torch.float32, device=dev)
    loss = loss_out if loss_out is not None else torch.empty(1, dtype=torch.float32, device=dev)
    td = torch.empty(rows, dtype=torch.float32, device=dev) if want_td else None
    nb = L.lib().mpg_q_loss_grad_workspace_bytes(ctypes.byref(cfg), L.c_int(rows))
    ws = workspace(dev, nb)
    L.call('mpg_q_loss_grad', ctypes.byref(cfg), L.ptr(_f32(q_params)), L.c_int(rows), L.ptr(_f32(obs)),
           L.ptr(_f32(act)), L.ptr(_f32(y)), L.c_float(inv_b_global if inv_b_global is not None else 1.0 / rows),
           L.ptr(loss), L.ptr(grad), L.ptr(td), L.ptr(ws), L.c_size_t(ws.numel()), L.stream())
    return loss, grad, td


def rollout_pg(cfg, policy_params, q1_params, obs0, eps, select, w, M=1, inv_b_global=None, all_steps_param_grad=False,
               grad_out=None, stats_out=None, n=None, noise_seed=0, noise_ctr=0):
    """mpg_rollout_pg: n-step model rollout + (mixed) policy gradient.  Returns (ret_sum, ret_sqsum, grad).
    eps=None draws the model noise inside the kernel (Philox(noise_seed, noise_ctr)); then pass n."""
    rows = obs0.shape[0]
    if eps is not None:
        n = eps.shape[0]
        assert eps.shape[1] == rows * M
    dev = obs0.device
    ns = len(select)
    grad = grad_out if grad_out is not None else torch.empty(policy_size(cfg), dtype=torch.float32, device=dev)
    stats = stats_out if stats_out is not None else torch.empty(2 * ns, dtype=torch.float32, device=dev)
    sel = (ctypes.c_int * ns)(*[int(k) for k in select])
    wv = (ctypes.c_float * ns)(*[float(x) for x in w])
    nb = L.lib().mpg_rollout_pg_workspace_bytes(ctypes.byref(cfg), L.c_int(rows), L.c_int(M), L.c_int(n), L.c_int(ns),
                                                L.c_int(int(all_steps_param_grad)))
    if nb == 0:
        raise L.MpgError('mpg_rollout_pg_workspace_bytes: unsupported configuration')
    ws = workspace(dev, nb, slot=1)
    L.call('mpg_rollout_pg', ctypes.byref(cfg), L.ptr(_f32(policy_params)), L.ptr(_f32(q1_params)), L.c_int(rows),
           L.c_int(M), L.c_int(n), sel, L.c_int(ns), wv, L.ptr(_f32(obs0)), L.ptr(_f32(eps) if eps is not None else None),
           L.c_u64(noise_seed), L.c_u64(noise_ctr), L.c_float(inv_b_global if inv_b_global is not None else 1.0 / rows), L.c_int(int(all_steps_param_grad)),
           L.ptr(stats[:ns]), L.ptr(stats[ns:]), L.ptr(grad), L.ptr(ws), L.c_size_t(ws.numel()), L.stream())
    return stats[:ns], stats[ns:], grad


CLIP_PARTS = 272          # MPG_CLIP_PARTS (include/mpg_hip.h)


def sq_partials(grad, seg_sizes, sq_part=None):
    ns = len(seg_sizes)
    part = sq_part if sq_part is not None else torch.empty(ns * CLIP_PARTS, dtype=torch.float32, device=grad.device)
    segs = (ctypes.c_int * ns)(*[int(s) for s in seg_sizes])
    L.call('mpg_sq_partials', L.ptr(_f32(grad)), segs, L.c_int(ns), L.ptr(part), L.stream())
    return part


def clip_adam_polyak(w, m, v, target, grad, sq_part, seg_sizes, clip, lr_t, do_adam, do_polyak, tau, norms, nonfinite=None,
                     wc_w=None, wc_target=None):
    """mpg_clip_adam_polyak: second half of the clip + Adam + Polyak in one launch"""
    ns = len(seg_sizes)
    segs = (ctypes.c_int * ns)(*[int(s) for s in seg_sizes])
    lr = (ctypes.c_float * ns)(*[float(x) for x in lr_t])
    da = (ctypes.c_int * ns)(*[int(x) for x in do_adam])
    dp = (ctypes.c_int * ns)(*[int(x) for x in do_polyak])
    L.call('mpg_clip_adam_polyak', L.ptr(_f32(w)), L.ptr(_f32(m)), L.ptr(_f32(v)), L.ptr(target), L.ptr(_f32(grad)),
           L.ptr(_f32(sq_part)), segs, L.c_int(ns), L.c_float(clip), lr, da, dp, L.c_float(tau), L.ptr(norms), L.ptr(nonfinite),
           _wc(wc_w), _wc(wc_target), L.stream())


def clip_by_global_norm(grad, seg_sizes, clip, norms_out=None, nonfinite=None, scratch=None):
    ns = len(seg_sizes)
    norms = norms_out if norms_out is not None else torch.empty(ns, dtype=torch.float32, device=grad.device)
    segs = (ctypes.c_int * ns)(*[int(s) for s in seg_sizes])
    L.call('mpg_clip_by_global_norm', L.ptr(_f32(grad)), segs, L.c_int(ns), L.c_float(clip), L.ptr(norms),
           L.ptr(nonfinite), L.ptr(scratch), L.stream())
    return norms


def adam_polyak(w, m, v, target, grad, seg_sizes, lr_t, do_adam, do_polyak, tau, skip_flag=None, wc_w=None, wc_target=None):
    ns = len(seg_sizes)
    segs = (ctypes.c_int * ns)(*[int(s) for s in seg_sizes])
    lr = (ctypes.c_float * ns)(*[float(x) for x in lr_t])
    da = (ctypes.c_int * ns)(*[int(x) for x in do_adam])
    dp = (ctypes.c_int * ns)(*[int(x) for x in do_polyak])
    L.call('mpg_adam_polyak', L.ptr(_f32(w)), L.ptr(_f32(m)), L.ptr(_f32(v)), L.ptr(target), L.ptr(_f32(grad)), segs,
           L.c_int(ns), lr, da, dp, L.c_float(tau), L.ptr(skip_flag),
           L.c_int(skip_flag.numel() if skip_flag is not None else 0), _wc(wc_w), _wc(wc_target), L.stream())


def rollout_q_target(cfg, policy_params, q1t, obs0, act0, eps, n=None, noise_seed=0, noise_ctr=0):
    rows = obs0.shape[0]
    n = eps.shape[0] if eps is not None else n
    y = torch.empty(rows, dtype=torch.float32, device=obs0.device)
    nb = L.lib().mpg_rollout_q_target_workspace_bytes(ctypes.byref(cfg), L.c_int(rows))
    ws = workspace(obs0.device, nb)
    L.call('mpg_rollout_q_target', ctypes.byref(cfg), L.ptr(_f32(policy_params)), L.ptr(_f32(q1t)), L.c_int(rows), L.c_int(n),
           L.ptr(_f32(obs0)), L.ptr(_f32(act0)), L.ptr(_f32(eps) if eps is not None else None), L.c_u64(noise_seed),
           L.c_u64(noise_ctr), L.ptr(y), L.ptr(ws), L.c_size_t(ws.numel()), L.stream())
    return y


def rollout_q_estimation(cfg, policy_params, q1t, obs0, act0, eps, select, M=1, noise_seed=0, noise_ctr=0):
    """mpg_rollout_q_estimation -> [n_select * rows] (the reference's concatenation of the selected slices)"""
    rows, ns = obs0.shape[0], len(select)
    y = torch.empty(ns * rows, dtype=torch.float32, device=obs0.device)
    sel = (ctypes.c_int * ns)(*[int(k) for k in select])
    nb = L.lib().mpg_rollout_q_estimation_workspace_bytes(ctypes.byref(cfg), L.c_int(rows), L.c_int(M), L.c_int(ns))
    if nb == 0:
        raise L.MpgError('mpg_rollout_q_estimation_workspace_bytes: unsupported configuration')
    ws = workspace(obs0.device, nb)
    L.call('mpg_rollout_q_estimation', ctypes.byref(cfg), L.ptr(_f32(policy_params)), L.ptr(_f32(q1t)), L.c_int(rows), L.c_int(M),
           sel, L.c_int(ns), L.ptr(_f32(obs0)), L.ptr(_f32(act0)), L.ptr(_f32(eps) if eps is not None else None),
           L.c_u64(noise_seed), L.c_u64(noise_ctr), L.ptr(y), L.ptr(ws), L.c_size_t(ws.numel()), L.stream())
    return y


def td3_policy_grad(cfg, policy_params, q1, q2, obs, inv_b_global=None, grad_out=None, stats_out=None):
    rows, dev = obs.shape[0], obs.device
    grad = grad_out if grad_out is not None else torch.empty(policy_size(cfg), dtype=torch.float32, device=dev)
    stats = stats_out if stats_out is not None else torch.empty(2, dtype=torch.float32, device=dev)
    nb = L.lib().mpg_td3_policy_grad_workspace_bytes(ctypes.byref(cfg), L.c_int(rows))
    ws = workspace(dev, nb, slot=1)
    L.call('mpg_td3_policy_grad', ctypes.byref(cfg), L.ptr(_f32(policy_params)), L.ptr(_f32(q1)), L.ptr(_f32(q2)),
           L.c_int(rows), L.ptr(_f32(obs)), L.c_float(inv_b_global if inv_b_global is not None else 1.0 / rows),
           L.ptr(stats[0:1]), L.ptr(stats[1:2]), L.ptr(grad), L.ptr(ws), L.c_size_t(ws.numel()), L.stream())
    return stats, grad


def mpg_gradients(cfg, n_q, params, target_params, obs, act, rew, obs_tp1, y_in, select, w, grad, stats, y_out, M=1, n=None,
                  eps=None, noise_seed=0, noise_ctr=0, inv_b_global=None, sq_part=None):
    """mpg_mpg_gradients: MPGLearner.compute_gradient without the clip (targets unless y_in, critic grads, mixed PG)."""
    rows, dev = obs.shape[0], obs.device
    if eps is not None:
        n = eps.shape[0]
    ns = len(select)
    sel = (ctypes.c_int * ns)(*[int(k) for k in select])
    wv = (ctypes.c_float * ns)(*[float(x) for x in w])
    nb = L.lib().mpg_mpg_gradients_workspace_bytes(ctypes.byref(cfg), L.c_int(rows), L.c_int(M), L.c_int(n), L.c_int(ns),
                                                   L.c_int(n_q))
    if nb == 0:
        raise L.MpgError('mpg_mpg_gradients_workspace_bytes: unsupported configuration')
    ws = workspace(dev, nb, slot=1)
    L.call('mpg_mpg_gradients', ctypes.byref(cfg), L.c_int(n_q), L.ptr(_f32(params)), L.ptr(target_params), L.c_int(rows),
           L.ptr(_f32(obs)), L.ptr(_f32(act)), L.ptr(rew), L.ptr(obs_tp1), L.ptr(y_in), L.c_int(M), L.c_int(n), sel,
           L.c_int(ns), wv, L.ptr(eps), L.c_u64(noise_seed), L.c_u64(noise_ctr),
           L.c_float(inv_b_global if inv_b_global is not None else 1.0 / rows), L.ptr(_f32(grad)), L.ptr(_f32(stats)),
           L.ptr(_f32(y_out)), L.ptr(sq_part), None, L.ptr(ws), L.c_size_t(ws.numel()), L.stream())
