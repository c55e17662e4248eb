"""SingleProcessOffPolicyOptimizer - mirror of optimizer.py:286-397: the per-iteration order
(sample every 10th iteration -> replay -> set_weights -> compute_gradient -> (priorities) -> NaN guard ->
apply_gradients) is the contract; Ray / TensorBoard plumbing is out of scope.  One instance per GPU process."""
import gc
import logging

logger = logging.getLogger(__name__)


def quiesce_gc():
    """Collect once, then move everything alive into the permanent generation (gc.freeze).  `import torch` leaves ~10^6
    container objects behind; a generation-2 pass over them takes 40-100 ms and the step loop's own small allocations
    trigger one every few thousand iterations - 10 % of a 0.45 ms step (tools/stall_scan.py).  After the freeze the
    collector only ever walks what the loop itself allocates.  Call it once after the training stack is built."""
    gc.collect()
    gc.freeze()


class SingleProcessOffPolicyOptimizer(object):
    def __init__(self, worker, learner, replay_buffer, evaluator, args, sampling_interval=10, fused=True,
                 always_exchange=False):
        self.args = args
        self.worker, self.learner, self.replay_buffer, self.evaluator = worker, learner, replay_buffer, evaluator
        self.num_sampled_steps = 0
        self.iteration = 0
        self.sampling_interval = sampling_interval
        self.stats = {}
        # single process: share one PolicyWithQs between worker and learner (no 1.6 MB weight copy per iteration)
        self.learner.share_policy(self.worker.policy_with_value)
        while not len(self.replay_buffer) >= self.args.replay_starts:      # optimizer.py:310-313
            sample_batch, count = self.worker.sample_with_count()
            self.num_sampled_steps += count
            self.replay_buffer.add_batch(sample_batch)
        # native step driver (mpg_step_begin/_end) when the stock MPG components are plugged in; otherwise the
        # method-by-method path below, which computes the same thing
        self._fused = None
        if fused:
            from .buffer import PrioritizedReplayBuffer
            from .learners import MPGLearner, NADPLearner, TD3Learner
            per = isinstance(replay_buffer, PrioritizedReplayBuffer)
            normal = getattr(args, 'buffer_type', 'normal') == 'normal'
            ok = (type(learner) is MPGLearner and not per and normal and not learner.deriv_interval_policy) or \
                 (type(learner) is NADPLearner and not per and normal and learner.num_batch_reuse == 1 and learner.n_q == learner.n_pi) or \
                 (type(learner) is TD3Learner and per == (not normal) and learner.num_batch_reuse == 1)
            if ok:
                from .fused import FusedMPGStep
                self._fused = FusedMPGStep(worker, learner, replay_buffer, sampling_interval, always_exchange=always_exchange)

    def set_profiler(self, prof):
        """attach an ops.Profiler (or None) to every cfg this optimizer launches with"""
        cfgs = [self.worker.policy_with_value.cfg, self.learner.policy_with_value.cfg]
        if self._fused is not None:
            cfgs.append(self._fused.c.cfg)
        old = getattr(self, '_prof', None)
        if old is not None:
            old.detach(*cfgs)
        for c in cfgs:
            c.prof = None
        self._prof = prof            # the optimizer keeps the timer alive for as long as its cfgs point at it
        if prof is not None:
            prof.attach(*cfgs)

    def get_stats(self):
        self.stats.update(dict(num_sampled_steps=self.num_sampled_steps, iteration=self.iteration))
        return self.stats

    def step(self):
        if self._fused is not None:
            if self.iteration % self.sampling_interval == 0:
                self.num_sampled_steps += self.worker.num_agent * self._fused.c.sample_iters
            self._fused.step(self.iteration)
            self.learner._lazy_stats = self.learner._native_lazy_stats(self.iteration)
            self.iteration += 1
            self._check_status()
            return
        if self.iteration % self.sampling_interval == 0:                   # optimizer.py:332-337
            sample_batch, count = self.worker.sample_with_count()
            self.num_sampled_steps += count
            self.replay_buffer.add_batch(sample_batch)
        samples = self.replay_buffer.replay()                              # :340-341
        self.learner.set_weights(self.worker.policy_with_value)            # :345 (shared object: no copy)
        grads = self.learner.compute_gradient(samples[:5], self.replay_buffer, samples[-1], self.iteration)   # :349
        if getattr(self.args, 'buffer_type', 'normal') == 'priority':     # :351-353
            info = self.learner.get_info_for_buffer()
            info['rb'].update_priorities(info['indexes'], info['td_error'])
        # NaN guard (:357-361) lives on the device: the clip kernel raises a flag that makes Adam see zero gradients
        self.worker.apply_gradients(self.iteration, self.learner.flat_grad)   # :362
        self.get_stats()
        self.iteration += 1
        self._check_status()

    def _check_status(self):
        """judge_is_nan / the engine's numerical envelope (include/mpg_hip.h): every kernel of the step ORs its findings into the
        shared policy's sticky status word; it is read (one host synchronisation) every `nan_check_interval` iterations, in BOTH
        branches of step() - the native driver never goes through worker.sample(), which is where the method path reads it
        (worker.py:95-107, optimizer.py:357-361 of the reference stop the run on the same conditions)."""
        k = getattr(self.worker, 'nan_check_interval', 100)
        if k > 0 and self.iteration % k == 0:
            self.worker.policy_with_value.check_status()

    def stop(self):
        pass
