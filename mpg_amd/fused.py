"""Native step driver binding (mpg_step_begin / mpg_step_end): one optimizer iteration enqueued from C++ so that the
Python interpreter is not on the launch path.  Built from - and kept in sync with - the stock worker / buffer / learner
objects, whose methods remain usable at any time: every fused step starts by reading their counters and buffers
(sync_in) and ends by writing them back (push)."""
import ctypes

import torch

from . import _lib as L
from . import dist as D
from .ops import CfgStruct

_P = ctypes.c_void_p


class GradOpts(ctypes.Structure):
    """mpg_grad_opts_t"""
    _fields_ = [('critics_ready_event', _P)]


class TrainCtx(ctypes.Structure):
    """mpg_train_ctx_t (include/mpg_hip.h)"""
    _fields_ = [
        ('cfg', CfgStruct), ('learner_version', ctypes.c_int), ('num_agent', ctypes.c_int), ('sample_iters', ctypes.c_int),
        ('sampling_interval', ctypes.c_int), ('batch', ctypes.c_int), ('n', ctypes.c_int), ('M', ctypes.c_int),
        ('n_select', ctypes.c_int), ('select', ctypes.c_int * 4), ('eta', ctypes.c_float), ('total_ite', ctypes.c_int),
        ('clip', ctypes.c_float), ('tau', ctypes.c_float), ('delay_update', ctypes.c_int), ('num_batch_reuse', ctypes.c_int),
        ('world_size', ctypes.c_int), ('grads_exchanged', ctypes.c_int), ('explore_sigma', ctypes.c_float), ('value_lr', ctypes.c_float * 3),
        ('policy_lr', ctypes.c_float * 3),
        ('worker_seed', ctypes.c_uint64), ('noise_ctr', ctypes.c_uint64), ('env_seed', ctypes.c_uint64),
        ('env_ctr', ctypes.c_uint64), ('replay_seed', ctypes.c_uint64), ('replay_times', ctypes.c_uint64),
        ('learner_seed', ctypes.c_uint64), ('learner_counter', ctypes.c_uint64),
        ('ring_capacity', ctypes.c_int), ('ring_next', ctypes.c_int), ('ring_size', ctypes.c_int),
        ('opt_steps', ctypes.c_longlong * 3),
        ('env_state', _P), ('w_obs', _P), ('w_act', _P), ('w_rew', _P), ('w_obs2', _P), ('w_done', _P), ('w_done_intended', _P),
        ('ring_obs', _P), ('ring_act', _P), ('ring_rew', _P), ('ring_obs2', _P), ('ring_done', _P),
        ('idx', _P), ('b_obs', _P), ('b_act', _P), ('b_rew', _P), ('b_obs2', _P), ('b_done', _P), ('b_targets', _P),
        ('params', _P), ('targets', _P), ('adam_m', _P), ('adam_v', _P), ('grad', _P), ('norms', _P), ('clip_scratch', _P), ('nonfinite', _P),
        ('l_env_state', _P), ('l_obs', _P), ('l_act', _P), ('l_rewards', _P), ('l_done', _P), ('l_done_intended', _P),
        ('ws0', _P), ('ws1', _P), ('ws0_bytes', ctypes.c_size_t), ('ws1_bytes', ctypes.c_size_t),
        ('smooth_sigma', ctypes.c_float), ('smooth_clip', ctypes.c_float), ('prioritized', ctypes.c_int),
        ('per_sum', _P), ('per_min', _P), ('per_stamp', _P), ('per_capacity', ctypes.c_int), ('per_max_priority', _P),
        ('per_alpha', ctypes.c_double), ('per_beta', ctypes.c_double), ('per_eps', ctypes.c_double),
        ('b_weights', _P), ('scratch', _P),
        ('critics_ready_event', _P), ('grad_opts', GradOpts), ('clip_partials_ready', ctypes.c_int)]


class FusedMPGStep(object):
    """step(iteration) == SingleProcessOffPolicyOptimizer.step for (OffPolicyWorker, replay buffer, learner) sharing one
    PolicyWithQs: MPGLearner + ReplayBuffer (learner_version 1 / 2), NADPLearner + ReplayBuffer (3), TD3Learner + ReplayBuffer
    or PrioritizedReplayBuffer (4; the priority update of optimizer.py:351-353 included)."""

    def __init__(self, worker, learner, rb, sampling_interval, always_exchange=False):
        from .buffer import PrioritizedReplayBuffer
        from .learners import MPGLearner, NADPLearner, TD3Learner
        per = isinstance(rb, PrioritizedReplayBuffer)
        assert per == (learner.args.buffer_type != 'normal')
        assert learner.policy_with_value is worker.policy_with_value
        self.worker, self.learner, self.rb = worker, learner, rb
        pw, a, dev = worker.policy_with_value, learner.args, worker.device
        self.pw = pw
        c = self.c = TrainCtx()
        c.cfg = pw.cfg
        if type(learner) is MPGLearner:
            assert not per
            c.learner_version = 1 if a.learner_version == 'MPG-v1' else 2
            sel = list(learner.num_rollout_list_for_policy_update)
            c.n, c.M, c.n_select = max(sel), learner.M, len(sel)
            for i, k in enumerate(sel):
                c.select[i] = k
        elif type(learner) is NADPLearner:
            assert not per and learner.n_q == learner.n_pi and learner.num_batch_reuse == 1
            c.learner_version, c.n, c.M, c.n_select = 3, learner.n_pi, 1, 2
        else:
            assert type(learner) is TD3Learner and learner.num_batch_reuse == 1
            c.learner_version, c.n, c.M, c.n_select = 4, 1, 1, 1
            c.smooth_sigma, c.smooth_clip = float(a.policy_smoothing_sigma), float(a.policy_smoothing_clip)
        c.num_agent, c.sample_iters = worker.num_agent, max(1, worker.batch_size // worker.num_agent)
        c.sampling_interval = sampling_interval
        c.batch = learner.batch_size
        c.eta, c.total_ite = float(getattr(a, 'eta', 0.1)), int(getattr(a, 'rule_based_bias_total_ite', 9000))
        c.clip, c.tau, c.delay_update = float(a.gradient_clip_norm), pw.tau, pw.delay_update
        c.num_batch_reuse, c.world_size = learner.num_batch_reuse, D.world_size()
        # always_exchange: run the collective (and the exchanged-gradient form of the clip) even in a one-process group
        self.always_exchange = bool(always_exchange)
        c.grads_exchanged = 1 if (c.world_size > 1 or self.always_exchange) else 0
        c.explore_sigma = float(worker.explore_sigma or 0.)
        for i in range(3):
            c.value_lr[i], c.policy_lr[i] = pw.schedules['Q1'][i], pw.schedules['policy'][i]
        c.worker_seed, c.env_seed, c.replay_seed, c.learner_seed = worker.seed, worker.env.seed, rb.seed, learner.seed
        n, B, od, ad = worker.num_agent, learner.batch_size, pw.obs_dim, pw.act_dim
        f = dict(dtype=torch.float32, device=dev)
        u8 = dict(dtype=torch.uint8, device=dev)
        self.t = t = dict(
            w_obs=worker.obs.clone(), w_act=torch.empty(n, ad, **f), w_rew=torch.empty(n, **f), w_obs2=torch.empty(n, od, **f),
            w_done=torch.ones(n, **u8), w_done_intended=torch.zeros(n, **u8),
            idx=torch.empty(B, dtype=torch.int32, device=dev), b_obs=torch.empty(B, od, **f), b_act=torch.empty(B, ad, **f),
            b_rew=torch.empty(B, **f), b_obs2=torch.empty(B, od, **f), b_done=torch.empty(B, **f), b_targets=torch.empty(B, **f))
        if c.learner_version == 1:
            t.update(l_obs=torch.empty(B, od, **f), l_act=torch.empty(B, ad, **f), l_rewards=torch.empty(c.n, B, **f),
                     l_done=torch.ones(B, **u8), l_done_intended=torch.zeros(B, **u8))
            c.l_env_state = L.ptr(learner.env._state)
        for k, v in t.items():
            setattr(c, k, L.ptr(v))
        c.env_state = L.ptr(worker.env._state)
        c.ring_capacity = rb._maxsize
        for k, v in (('ring_obs', rb.obs), ('ring_act', rb.act), ('ring_rew', rb.rew), ('ring_obs2', rb.obs2), ('ring_done', rb.done)):
            setattr(c, k, L.ptr(v))
        c.params, c.targets, c.adam_m, c.adam_v = L.ptr(pw.params), L.ptr(pw.targets), L.ptr(pw.m), L.ptr(pw.v)
        c.grad, c.norms, c.nonfinite = L.ptr(learner.flat), L.ptr(learner.norms), L.ptr(pw.nonfinite)
        c.clip_scratch = L.ptr(learner.clip_scratch)
        if c.learner_version == 4:
            self.scratch = torch.empty(max(B * (ad + 3), 2 * n) + 64, **f)
            c.scratch = L.ptr(self.scratch)
            c.prioritized = 1 if per else 0
            if per:
                t['b_weights'] = torch.empty(B, **f)
                c.b_weights = L.ptr(t['b_weights'])
                c.per_sum, c.per_min, c.per_stamp = L.ptr(rb._it_sum), L.ptr(rb._it_min), L.ptr(rb._stamp)
                c.per_capacity, c.per_max_priority = rb._cap, L.ptr(rb._max_priority)
                c.per_alpha, c.per_beta, c.per_eps = rb._alpha, rb._beta, rb._eps
        # scheduling option of the MPG step (include/mpg_hip.h, mpg_grad_opts_t; leaves every number unchanged): with an exchange,
        # MPG_OVERLAP_EXCHANGE=1 (off by default): the critics' gradient is finished ahead of the reverse sweep and exchanged on a
        # second stream under it (costs ~9 us on one GPU, where there is nothing to hide)
        import os
        self.overlap = None
        if c.learner_version in (1, 2):
            if c.grads_exchanged and os.environ.get('MPG_OVERLAP_EXCHANGE') == '1':
                side = torch.cuda.Stream(device=dev)
                e1, e2 = torch.cuda.Event(), torch.cuda.Event()
                e1.record()                       # (the handle exists once the event has been recorded)
                e2.record()
                n_crit = int(sum(pw.sizes[:len(pw.names) - 1]))
                self.overlap = (side, e1, e2, n_crit)
                c.critics_ready_event = e1.cuda_event
        # round 6, one-shot backend without the overlap option: the step writes its partial gradient STRAIGHT into this rank's staging
        # slot of the exchange (no copy into it) and the exchange's sum leaves the clip's partials of the reduced gradient behind
        # (mpg_sum_slots_sq; mpg_train_ctx_t.clip_partials_ready) - on one GPU the exchange path is then one launch on top of the
        # unexchanged step instead of three
        self.slot_exchange = bool(c.grads_exchanged) and self.overlap is None and D._exchange == 'oneshot' and \
            os.environ.get('MPG_SLOT_EXCHANGE', '1') != '0'
        w0, w1 = ctypes.c_size_t(0), ctypes.c_size_t(0)
        L.call('mpg_step_workspace_bytes', ctypes.byref(c), ctypes.byref(w0), ctypes.byref(w1))
        self.ws0 = torch.empty(w0.value + 256, dtype=torch.uint8, device=dev)
        self.ws1 = torch.empty(w1.value + 256, dtype=torch.uint8, device=dev)
        c.ws0, c.ws1, c.ws0_bytes, c.ws1_bytes = L.ptr(self.ws0), L.ptr(self.ws1), self.ws0.numel(), self.ws1.numel()
        self._lib = L.lib()
        self._ref = ctypes.byref(c)
        # the python objects now look at the driver's buffers
        learner.batch_data = {'batch_obs': t['b_obs'], 'batch_actions': t['b_act'], 'batch_rewards': t['b_rew'],
                              'batch_obs_tp1': t['b_obs2'], 'batch_dones': t['b_done'], 'batch_targets': t['b_targets']}
        learner._views = None
        self.pull()

    def pull(self):
        """python objects -> context counters"""
        c, w, rb, ln, pw = self.c, self.worker, self.rb, self.learner, self.pw
        c.noise_ctr, c.env_ctr, c.replay_times, c.learner_counter = w._noise_ctr, w.env._ctr, rb.replay_times, ln.counter
        c.worker_seed, c.env_seed, c.replay_seed, c.learner_seed = w.seed, w.env.seed, rb.seed, ln.seed
        c.ring_next, c.ring_size = rb._next_idx, rb._size
        for i, n in enumerate(pw.names):
            c.opt_steps[i] = pw.opt_steps[n]

    def push(self):
        """context counters -> python objects (so their own methods can be mixed with fused steps)"""
        c, w, rb, ln, pw = self.c, self.worker, self.rb, self.learner, self.pw
        w._noise_ctr, w.env._ctr, rb.replay_times, ln.counter = c.noise_ctr, c.env_ctr, c.replay_times, c.learner_counter
        rb._next_idx, rb._size = c.ring_next, c.ring_size
        for i, n in enumerate(pw.names):
            pw.opt_steps[n] = c.opt_steps[i]
        w.obs = w.env.obs = self.t['w_obs']
        w.env.done = self.t['w_done']
        w.env._initialised = True

    def sync_in(self):
        """python objects -> driver (counters, and the worker's observation / done buffers if they were re-bound):
        whatever the stock methods did since the last fused step - worker.sample(), rb.add_batch(), rb.replay(),
        learner.compute_gradient(), a checkpoint restore - is what the next fused step continues from."""
        w = self.worker
        if w.obs is not self.t['w_obs']:
            self.t['w_obs'].copy_(w.obs)
        if w.env.done is not self.t['w_done']:
            self.t['w_done'].copy_(w.env.done)
        self.pull()

    def reload(self):
        """after the python objects were restored from a checkpoint: refresh the driver's own buffers and counters"""
        self.sync_in()
        self.push()

    def step(self, iteration):
        self.sync_in()           # a few host scalar copies; the tensor copies only happen after a stock-method call
        s = L.stream()
        flat = self.learner.flat
        slot = D.grad_slot(flat.numel(), flat.device, force=self.always_exchange) if self.slot_exchange else None
        if slot is not None:
            self.c.grad = slot.data_ptr()
        L.check(self._lib.mpg_step_begin(self._ref, ctypes.c_int(iteration), s), 'mpg_step_begin')
        if self.c.grads_exchanged:
            # the ONE exchange step, timed under the caller's kernel timer (slot 8, HIP events on the launch stream: bench.py's
            # `exchange_ms`) - a null or stopped timer makes both calls no-ops
            prof = ctypes.c_void_p(self.c.cfg.prof)
            self._lib.mpg_prof_region_begin(prof, ctypes.c_int(8), s)
            if self.overlap is not None:
                # the critics' slice was complete when the library recorded e1 (ahead of the reverse sweep): its exchange runs on
                # the side stream under the sweep, the policy's slice + statistics follow the sweep on the launch stream
                side, e1, e2, n_crit = self.overlap
                main = torch.cuda.current_stream()
                side.wait_event(e1)
                with torch.cuda.stream(side):
                    D.all_reduce_sum_(flat[:n_crit], force=self.always_exchange, tag=1)
                    e2.record(side)
                D.all_reduce_sum_(flat[n_crit:], force=self.always_exchange)
                main.wait_event(e2)
            elif slot is not None:
                D.all_reduce_sum_(flat, force=self.always_exchange, in_slot=True, seg_sizes=self.pw.sizes, sq_part=self.learner.clip_scratch)
                self.c.grad, self.c.clip_partials_ready = flat.data_ptr(), 1
            else:
                D.all_reduce_sum_(flat, force=self.always_exchange)
            self._lib.mpg_prof_region_end(prof, ctypes.c_int(8), s)
        L.check(self._lib.mpg_step_end(self._ref, ctypes.c_int(iteration), s), 'mpg_step_end')
        self.push()
