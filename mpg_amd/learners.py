"""Learners - device mirrors of learners/mpg_learner.py (MPGLearner), learners/nadp.py (NADPLearner) and
learners/td3.py (TD3Learner): same constructor signature `(policy_cls, args)`, same methods the optimizer calls
(`set_weights`, `compute_gradient(batch5, rb, indexes, iteration)`, `get_stats`, `get_info_for_buffer`), same
output order `q1 (+q2) + policy`.

Data-parallel form: each process computes sum-reduced, UN-clipped gradient partials already scaled
by 1/B_global into one flat buffer [grads | stats]; ONE all-reduce; then per-network tf.clip_by_global_norm
(non-linear, must follow the reduce).  On one process this is exactly the reference computation."""
import numpy as np
import torch

from . import dist as D
from . import ops
from .envs import PathTrackingEnv


def rule_based_weights(ite, total_ite, eta, select):
    """MPGLearner.rule_based_weights, mpg_learner.py:384-399 (float32 like the TF graph)."""
    f = np.float32
    lam = f(1. - eta) + f(2. * eta / total_ite) * f(ite)
    lam = f(min(max(lam, f(0.)), f(1.5)))
    if lam < 1.:
        biases = np.array([np.power(lam, f(i)) for i in select], dtype=f)
    else:
        mx = max(select)
        biases = np.array([np.power(f(2.) - lam, f(mx - i)) for i in select], dtype=f)
    inv = (f(1.) / (biases + f(1e-8))).astype(f)
    e = np.exp(inv - inv.max()).astype(f)
    return (e / e.sum()).astype(f)


N_STATS = 16     # floats appended to the gradient buffer and summed by the same all-reduce


class _LearnerBase(object):
    def __init__(self, policy_cls, args, device='cuda'):
        self.args = args
        self.device = torch.device(device)
        self.batch_size = args.replay_batch_size
        self.policy_with_value = policy_cls(**vars(args), device=device)
        self.cfg = self.policy_with_value.cfg
        self.batch_data = {}
        self.counter = 0
        self.num_batch_reuse = getattr(args, 'num_batch_reuse', 1)
        self.stats = {}
        self.info_for_buffer = {}
        pw = self.policy_with_value
        self.n_grad = int(pw.offsets[-1])
        self.flat = torch.zeros(self.n_grad + N_STATS, dtype=torch.float32, device=self.device)
        self.norms = torch.zeros(len(pw.names), dtype=torch.float32, device=self.device)
        self.clip_scratch = torch.zeros(len(pw.names) * ops.CLIP_PARTS, dtype=torch.float32, device=self.device)
        self.seed = int(getattr(args, 'seed', 0)) + 12345
        self._noise_gen = torch.Generator(device=self.device)
        self._noise_gen.manual_seed(self.seed)
        self._views = None
        self._lazy_stats = None

    # ---- optimizer-facing API ----
    def get_stats(self):
        """Host copy of the stats of the last compute_gradient (device scalars are only read here, so the training
        loop itself never synchronises)."""
        if self._lazy_stats is not None:
            self.stats.update(self._lazy_stats())
        return {k: (v.item() if isinstance(v, torch.Tensor) and v.numel() == 1 else
                    (v.tolist() if isinstance(v, torch.Tensor) else v)) for k, v in self.stats.items()}

    def get_info_for_buffer(self):
        return self.info_for_buffer

    def get_weights(self):
        return self.policy_with_value.get_weights()

    def set_weights(self, weights):
        if weights is self.policy_with_value or weights is None:
            return
        return self.policy_with_value.set_weights(weights)

    def share_policy(self, policy):
        """Single-process mode: learner and worker use ONE PolicyWithQs instead of copying 1.6 MB of weights every
        iteration (optimizer.py:345 `learner.set_weights(worker.get_weights())` becomes a no-op)."""
        self.policy_with_value = policy
        self.cfg = policy.cfg
        self._views = None

    def set_ppc_params(self, params):
        pass

    def grad(self, name):
        pw = self.policy_with_value
        i = pw.names.index(name)
        return self.flat[pw.offsets[i]:pw.offsets[i + 1]]

    def _get_batch(self, batch_data):
        def f32(t):
            if t.dtype == torch.float32 and t.device == self.device and t.is_contiguous():
                return t
            return t.to(self.device, torch.float32).contiguous()
        self.batch_data = {k: f32(batch_data[i])
                           for i, k in enumerate(('batch_obs', 'batch_actions', 'batch_rewards', 'batch_obs_tp1', 'batch_dones'))}

    def _finish(self, iteration, clip):
        """all-reduce, clip per network, expose the reference's list view."""
        pw = self.policy_with_value
        D.all_reduce_sum_(self.flat)
        if self._views is None:
            self.flat_grad = self.flat[:self.n_grad]
            self._views = []
            for i, n in enumerate(pw.names):
                self._views += pw._as_list(self.flat[pw.offsets[i]:pw.offsets[i + 1]], n)
        ops.clip_by_global_norm(self.flat_grad, pw.sizes, clip, norms_out=self.norms, nonfinite=pw.nonfinite,
                                scratch=self.clip_scratch)
        self.stats['iteration'] = iteration
        return self._views


class MPGLearner(_LearnerBase):
    def __init__(self, policy_cls, args, device='cuda'):
        super().__init__(policy_cls, args, device)
        self.sample_num_in_learner = args.sample_num_in_learner
        self.M = args.M
        self.num_rollout_list_for_policy_update = list(args.num_rollout_list_for_policy_update)
        self.deriv_interval_policy = bool(getattr(args, 'deriv_interval_policy', False))   # mpg_learner.py:247-248
        self.env = None
        if args.learner_version == 'MPG-v1':
            self.env = PathTrackingEnv(num_agent=self.batch_size, num_future_data=args.num_future_data, device=device)

    # ---- heuristic-bias rollout (defined but never called by the reference's compute_gradient either) ----
    def model_rollout_for_q_estimation(self, start_obses, start_actions, eps=None):
        """mpg_learner.py:180-224: from (s, a_replay) roll the model, later actions from pi_theta, bootstrap every selected
        slice of args.num_rollout_list_for_q_estimation with Q1_target, mean over the M copies; returns the selected
        slices concatenated ([len(list) * B], no gradient).  eps: optional [max(list)][M*B] standard-normal model noise
        (default: Philox draws keyed by the learner's seed and call counter)."""
        sel = list(getattr(self.args, 'num_rollout_list_for_q_estimation', []) or [])
        assert sel, 'args.num_rollout_list_for_q_estimation is empty'
        pw = self.policy_with_value
        self._qest_calls = getattr(self, '_qest_calls', 0) + 1
        return ops.rollout_q_estimation(self.cfg, pw.net('policy'), pw.net('Q1', True), start_obses, start_actions, eps, sel,
                                        M=self.M, noise_seed=self.seed + 7, noise_ctr=self._qest_calls)

    # ---- targets ----
    def compute_clipped_double_q_target(self):
        """mpg_learner.py:126-134"""
        pw, b = self.policy_with_value, self.batch_data
        return ops.q_targets(self.cfg, pw.net('policy', True), pw.net('Q1', True), pw.net('Q2', True),
                             b['batch_rewards'], b['batch_obs_tp1'])

    def sample(self, start_obs, start_action):
        """mpg_learner.py:109-124: n real-env steps; first action from replay, then the ONLINE policy, no noise."""
        pw = self.policy_with_value
        obs = start_obs
        self.env.reset(init_obs=obs)
        rewards = []
        for t in range(self.sample_num_in_learner):
            action = start_action if t == 0 else ops.policy_action(self.cfg, pw.net('policy'), obs)
            obs, r, _, _ = self.env.step(action)
            rewards.append(r)
        return {'all_rewards': torch.stack(rewards).contiguous(), 'last_obs': obs}

    def compute_n_step_target(self):
        """mpg_learner.py:146-169"""
        pw, b = self.policy_with_value, self.batch_data
        ro = self.sample(b['batch_obs'], b['batch_actions'])
        return ops.nstep_targets(self.cfg, pw.net('policy', True), pw.net('Q1', True), ro['all_rewards'], ro['last_obs'])

    def compute_td_error(self):
        """mpg_learner.py:136-144 (signed)."""
        pw, b = self.policy_with_value, self.batch_data
        y1 = ops.q_targets(self.cfg, pw.net('policy', True), pw.net('Q1', True), None, b['batch_rewards'], b['batch_obs_tp1'])
        return y1 - pw.compute_Q1(b['batch_obs'], b['batch_actions'])

    def get_batch_data(self, batch_data, rb, indexes):
        self._get_batch(batch_data)
        if self.args.learner_version == 'MPG-v1':
            target = self.compute_n_step_target()
        elif self.args.learner_version == 'MPG-v2':
            target = self.compute_clipped_double_q_target()
        else:
            raise ValueError(self.args.learner_version)
        self.batch_data['batch_targets'] = target
        if self.args.buffer_type != 'normal':
            self.info_for_buffer.update(dict(td_error=self.compute_td_error(), rb=rb, indexes=indexes))

    def draw_model_noise(self, n, cols):
        return torch.randn(n, cols, generator=self._noise_gen, device=self.device, dtype=torch.float32)

    def compute_gradient(self, batch_data, rb, indexes, iteration, eps=None):
        """mpg_learner.py:401-455.  Returns the list [q1 (6 arrays) (+ q2) + policy (6 arrays)] of device tensors
        (views of one flat buffer, also available as `self.flat_grad`).  eps: optional [n, M*B] standard-normal model
        noise (parity tests); by default it is drawn inside the rollout kernel."""
        if self.counter % self.num_batch_reuse == 0:
            self.get_batch_data(batch_data, rb, indexes)
        self.counter += 1
        pw, b = self.policy_with_value, self.batch_data
        rows = b['batch_obs'].shape[0]
        world = D.world_size()
        inv_b = 1.0 / (rows * world)
        select = self.num_rollout_list_for_policy_update
        ws = rule_based_weights(iteration, self.args.rule_based_bias_total_ite, self.args.eta, select)
        if self.deriv_interval_policy:
            # every rollout step goes through pi_theta (full BPTT, mpg_learner.py:247-248): the fine-grained entry points
            stats = self.flat[self.n_grad:]
            for i, nm in enumerate(n for n in pw.names if n != 'policy'):
                ops.q_loss_grad(self.cfg, pw.net(nm), b['batch_obs'], b['batch_actions'], b['batch_targets'],
                                inv_b_global=inv_b, grad_out=self.grad(nm), loss_out=stats[i:i + 1])
            ops.rollout_pg(self.cfg, pw.net('policy'), pw.net('Q1'), b['batch_obs'], eps, select, ws, M=self.M,
                           inv_b_global=inv_b, all_steps_param_grad=True, grad_out=self.grad('policy'),
                           stats_out=stats[2:2 + 2 * len(select)], n=max(select), noise_seed=self.seed, noise_ctr=self.counter)
            out = self._finish(iteration, float(self.args.gradient_clip_norm))
            self._lazy_stats = self._mpg_lazy_stats(iteration)
            return out
        # one native call: critic losses/gradients + model rollout + mixed policy gradient (5 launches); the targets
        # were computed by get_batch_data (the reference caches them per batch, mpg_learner.py:402-403)
        ops.mpg_gradients(self.cfg, len(pw.names) - 1, pw.params, pw.targets, b['batch_obs'], b['batch_actions'],
                          b['batch_rewards'], b['batch_obs_tp1'], b['batch_targets'], select, ws, self.flat[:self.n_grad],
                          self.flat[self.n_grad:], b['batch_targets'], M=self.M, n=max(select), eps=eps, noise_seed=self.seed,
                          noise_ctr=self.counter, inv_b_global=inv_b)
        out = self._finish(iteration, float(self.args.gradient_clip_norm))
        self._lazy_stats = self._mpg_lazy_stats(iteration)
        return out

    def _native_lazy_stats(self, iteration):
        """what the native step driver leaves in the statistics slots, as get_stats() reports it"""
        return self._mpg_lazy_stats(iteration)

    def _mpg_lazy_stats(self, iteration):
        """stats of mpg_learner.py:433-452, evaluated only when get_stats() is called"""
        pw = self.policy_with_value
        select = self.num_rollout_list_for_policy_update
        ns, nq = len(select), len(pw.names) - 1
        stats = self.flat[self.n_grad:]
        B = self.batch_size * D.world_size()

        def lazy():
            ws = rule_based_weights(iteration, self.args.rule_based_bias_total_ite, self.args.eta, select)
            mean_ret = stats[2:2 + ns] / B
            d = dict(iteration=iteration, value_mean=mean_ret[select.index(0)] if 0 in select else None,
                     policy_total_loss=-(torch.as_tensor(ws, device=self.device) * mean_ret).sum(),
                     policy_gradient_norm=self.norms[nq], q_loss1=stats[0], q_gradient_norm1=self.norms[0],
                     num_rollout_list=select, w_list=list(map(float, ws)), all_losses=-mean_ret)
            if nq == 2:
                d.update(q_loss2=stats[1], q_gradient_norm2=self.norms[1])
            return d
        return lazy


class NADPLearner(_LearnerBase):
    """n-step ADP (learners/nadp.py:23-241), config 3: the Q target AND the policy loss come from 25-step MODEL rollouts;
    every rollout step goes through pi_theta, so parameter gradients accumulate at all 26 policy evaluations."""

    def __init__(self, policy_cls, args, device='cuda'):
        super().__init__(policy_cls, args, device)
        self.M = args.M
        assert self.M == 1
        self.n_q = max(args.num_rollout_list_for_q_estimation)
        self.n_pi = args.num_rollout_list_for_policy_update[0]

    def get_batch_data(self, batch_data, rb, indexes):
        self._get_batch(batch_data)

    def compute_gradient(self, batch_data, rb, indexes, iteration, eps_q=None, eps_pi=None):
        """nadp.py:209-241"""
        if self.counter % self.num_batch_reuse == 0:
            self.get_batch_data(batch_data, rb, indexes)
        self.counter += 1
        pw, b = self.policy_with_value, self.batch_data
        rows = b['batch_obs'].shape[0]
        world = D.world_size()
        inv_b = 1.0 / (rows * world)
        stats = self.flat[self.n_grad:]
        targets = ops.rollout_q_target(self.cfg, pw.net('policy'), pw.net('Q1', True), b['batch_obs'], b['batch_actions'],
                                       eps_q, n=self.n_q, noise_seed=self.seed, noise_ctr=2 * self.counter)   # nadp.py:87-126
        self.batch_data['batch_targets'] = targets
        ops.q_loss_grad(self.cfg, pw.net('Q1'), b['batch_obs'], b['batch_actions'], targets, inv_b_global=inv_b,
                        grad_out=self.grad('Q1'), loss_out=stats[0:1])                               # :173-184
        # slice 0 only feeds value_mean (weight 0); the loss is -R_n (nadp.py:168-171)
        ops.rollout_pg(self.cfg, pw.net('policy'), pw.net('Q1'), b['batch_obs'], eps_pi, [0, self.n_pi], [0.0, 1.0], M=1,
                       inv_b_global=inv_b, all_steps_param_grad=True, grad_out=self.grad('policy'), stats_out=stats[2:6],
                       n=self.n_pi, noise_seed=self.seed, noise_ctr=2 * self.counter + 1)
        out = self._finish(iteration, float(self.args.gradient_clip_norm))
        self._lazy_stats = self._native_lazy_stats(iteration)
        return out

    def _native_lazy_stats(self, iteration):
        """evaluated only when get_stats() is called: no elementwise launches in the training loop"""
        stats, B = self.flat[self.n_grad:], self.batch_size * D.world_size()
        return lambda: dict(q_loss=stats[0], policy_loss=-stats[3] / B, value_mean=stats[2] / B,
                            q_gradient_norm=self.norms[0], policy_gradient_norm=self.norms[1])


class TD3Learner(_LearnerBase):
    """learners/td3.py:22-188, config 4."""

    def compute_clipped_double_q_target(self, smooth_eps=None):
        """td3.py:69-81"""
        pw, b = self.policy_with_value, self.batch_data
        rows = b['batch_obs'].shape[0]
        if smooth_eps is None:
            smooth_eps = self._smoothing_noise(rows)
        return ops.q_targets(self.cfg, pw.net('policy', True), pw.net('Q1', True), pw.net('Q2', True), b['batch_rewards'],
                             b['batch_obs_tp1'], smooth_eps=smooth_eps, smooth_sigma=self.args.policy_smoothing_sigma,
                             smooth_clip=self.args.policy_smoothing_clip)

    def _smoothing_noise(self, rows):
        """target-policy smoothing noise (td3.py:74, tf.random.normal in the reference): the library's Philox stream keyed by
        (learner seed, the gradient step this batch is fetched for) - the numbers the native step driver draws"""
        return ops.normal_fill(rows * self.cfg.act_dim, self.seed, self.counter + 1, self.device).view(rows, self.cfg.act_dim)

    def compute_td_error(self):
        """td3.py:83-92 (signed)."""
        pw, b = self.policy_with_value, self.batch_data
        y1 = ops.q_targets(self.cfg, pw.net('policy', True), pw.net('Q1', True), None, b['batch_rewards'], b['batch_obs_tp1'])
        return y1 - pw.compute_Q1(b['batch_obs'], b['batch_actions'])

    def get_batch_data(self, batch_data, rb, indexes, smooth_eps=None):
        self._get_batch(batch_data)
        self._y1 = None
        if self.args.buffer_type != 'normal' and self.num_batch_reuse == 1:
            # the priorities' td error y1 - Q1(s, a) (td3.py:83-92): y1 shares the target policy's pass with the clipped double-Q
            # target (mpg_td3_targets: pi_t(s') once, not twice), and Q1 on the batch is what the critic-loss pass of
            # compute_gradient evaluates anyway, on the same weights: keep y1 and finish there (two network passes less)
            pw, b = self.policy_with_value, self.batch_data
            rows = b['batch_obs'].shape[0]
            if smooth_eps is None:
                smooth_eps = self._smoothing_noise(rows)
            self.batch_data['batch_targets'], self._y1 = ops.td3_targets(
                self.cfg, pw.net('policy', True), pw.net('Q1', True), pw.net('Q2', True), b['batch_rewards'], b['batch_obs_tp1'],
                smooth_eps, smooth_sigma=self.args.policy_smoothing_sigma, smooth_clip=self.args.policy_smoothing_clip)
            self.info_for_buffer.update(dict(td_error=None, rb=rb, indexes=indexes))
            return
        self.batch_data['batch_targets'] = self.compute_clipped_double_q_target(smooth_eps)
        if self.args.buffer_type != 'normal':
            self.info_for_buffer.update(dict(td_error=self.compute_td_error(), rb=rb, indexes=indexes))

    def compute_gradient(self, batch_data, rb, indexes, iteration, smooth_eps=None):
        """td3.py:150-188"""
        if self.counter % self.num_batch_reuse == 0:
            self.get_batch_data(batch_data, rb, indexes, smooth_eps)
        self.counter += 1
        pw, b = self.policy_with_value, self.batch_data
        rows = b['batch_obs'].shape[0]
        world = D.world_size()
        inv_b = 1.0 / (rows * world)
        stats = self.flat[self.n_grad:]
        for i, nm in enumerate(('Q1', 'Q2')):
            pending = nm == 'Q1' and getattr(self, '_y1', None) is not None
            td = ops.q_loss_grad(self.cfg, pw.net(nm), b['batch_obs'], b['batch_actions'], b['batch_targets'], inv_b_global=inv_b,
                                 grad_out=self.grad(nm), loss_out=stats[i:i + 1], want_td=pending)[2]
            if pending:        # td = Q1(s, a) - y  =>  y1 - Q1(s, a) = (y1 - y) - td  (one launch, like the native step driver)
                self.info_for_buffer['td_error'] = ops.td3_priority_errors(self._y1, b['batch_targets'], td)
                self._y1 = None
        ops.td3_policy_grad(self.cfg, pw.net('policy'), pw.net('Q1'), pw.net('Q2'), b['batch_obs'], inv_b_global=inv_b,
                            grad_out=self.grad('policy'), stats_out=stats[2:4])
        out = self._finish(iteration, float(self.args.gradient_clip_norm))
        self._lazy_stats = self._native_lazy_stats(iteration)
        return out

    def _native_lazy_stats(self, iteration):
        stats, B = self.flat[self.n_grad:], self.batch_size * D.world_size()

        def lazy():        # evaluated only when get_stats() is called: no elementwise launches in the training loop
            mean = stats[2] / B
            return dict(q_loss1=stats[0], q_loss2=stats[1], policy_loss=-mean, value_mean=mean,
                        value_var=stats[3] / B - mean * mean, q_gradient_norm1=self.norms[0],
                        q_gradient_norm2=self.norms[1], policy_gradient_norm=self.norms[2])
        return lazy
