/*
 * mpg_hip.h - C ABI of libmpg_hip.so, the MI355X (gfx950) drop-in for the data-parallel hot path of
 * idthanm/mpg (Mixed Policy Gradient).
 *
 * The reference has no FFI: its "plugin API" is a set of duck-typed Python classes picked through
 * registries (train_scripts/train_script.py:39-51, envs_and_models/__init__.py:13-15).  Each entry point
 * below replaces the arithmetic of one group of those methods; the file:line it replaces is cited on
 * every declaration (paths relative to the reference root).  The python package mpg_amd keeps the reference's class and
 * method names on top of this ABI; INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. a torch tensor), contiguous float32
 *     unless stated; nothing is allocated, freed or copied to the host behind the caller's back;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is enqueued on it and
 *     the call returns immediately (no host synchronisation);
 *   - return value: 0 on success, a negative MPG_E* code or -(hipError_t) otherwise; no exceptions cross
 *     the boundary;
 *   - re-entrant per (device, stream): no global mutable state (caches and timers are caller-owned handles, see
 *     mpg_wcache_t / mpg_prof_t; hyper-parameters arrive in mpg_cfg_t, never through the environment);
 *   - networks are 2-hidden-layer ELU MLPs with 256 hidden units (model.py:20-43), parameters stored as
 *     ONE flat float32 vector in Keras order  W1[in][256] b1[256] W2[256][256] b2[256] W3[256][out] b3[out]
 *     (kernels row-major (in,out), exactly `Model.get_weights()` flattened);
 *   - row batches (obs, act, ...) are row-major [rows][dim] like the reference's numpy arrays.
 */
#ifndef MPG_HIP_H
#define MPG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mpg_stream_t; /* hipStream_t */

enum {
    MPG_OK = 0,
    MPG_EINVAL = -1000,   /* bad argument (null pointer, unsupported dimension, ...) */
    MPG_EWORKSPACE = -1001 /* workspace too small: call the *_workspace_bytes query */
};

enum { MPG_ENV_PATH_TRACKING = 0, MPG_ENV_INVERTED_PENDULUM = 1 };
enum { MPG_ACT_LINEAR = 0, MPG_ACT_TANH = 1 };
enum { MPG_HIDDEN = 256 };
/* floats per agent of the opaque env state block, stored SoA [MPG_ENV_STATE_DIM][n]:
 * v_x, v_y, r, y, phi, x (veh_full_state) + delta_y, delta_phi (the two derived entries of veh_state). */
enum { MPG_ENV_STATE_DIM = 8 };

/* ABI version of this header; bumped on any signature change. */
int mpg_abi_version(void);
/* Human-readable description of the last error on this thread ("" if none). */
const char* mpg_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Caller-owned handles (no reference counterpart).  The library keeps NO process-wide mutable state: the packed
 * weight images and the optional kernel timer below are objects the caller creates, owns and passes in (through
 * mpg_cfg_t or an explicit argument).  The one thing kept on the library side is the thread-local text behind
 * mpg_last_error().
 * ---------------------------------------------------------------------------------------------- */

/* Packed register images of the 256x256 hidden kernels ("weight cache", acceleration only).  Describes ONE flat
 * parameter vector `params` = [net0 | net1 | ...] (Keras order each) and a caller-owned device array `packed` of
 * mpg_weight_cache_floats(n_nets) floats holding, per network, the forward and the transposed image of W2 in the
 * order the lanes of the MLP engine consume them (1 KiB coalesced loads instead of strided ones).  Entry points that
 * are told about a descriptor (mpg_cfg_t.wcache, or their explicit `wcache` argument) and receive a pointer into
 * `params` use the images; results are bit-identical with and without them.  The descriptor itself is HOST memory,
 * read at call time.  Whoever writes `params` keeps `packed` current: mpg_adam_polyak / mpg_clip_adam_polyak do so
 * themselves for the descriptors they are given, anything else calls mpg_weight_cache_pack afterwards. */
typedef struct {
    const float* params;  /* device: base of the flat vector the images mirror */
    float* packed;        /* device: mpg_weight_cache_floats(n_nets) floats */
    int n_nets;           /* <= 8 */
    int in_dim[8], out_dim[8];
    int* status;          /* device, nullable: MPG_STATUS_* bits are OR-ed into it (see "Numerical envelope" below) */
} mpg_wcache_t;
size_t mpg_weight_cache_floats(int n_nets);
/* (re)builds both images of every network of `wc` from wc->params: one launch on `stream`. */
int mpg_weight_cache_pack(const mpg_wcache_t* wc, mpg_stream_t stream);

/* Numerical envelope of the hidden-layer engine and how leaving it is reported.  The 256 x 256 hidden-layer products run
 * on the f16 matrix pipe with every float32 operand split into two fp16 halves (hi + lo: 22-23 significant bits, exact
 * products, float32 accumulation; csrc/mlp_core.h).  Operands are pre-scaled by powers of two; what that leaves as the
 * supported range is
 *     |first hidden activation| < 4094           (enters the image as x * 16; fp16 ends at 65504)
 *     |any network parameter|   < 1023.5         (hidden kernels enter as W * 64; the bound also keeps the reverse layer's
 *                                                 row-scaled operands, <= 16 * out * |W3|, in range)
 * Gradients entering the reverse layer and the weight-gradient product are scaled by data-dependent powers of two (per row
 * / per chunk of rows) and have no envelope of their own.  Outside the range a result would silently be wrong (an fp16
 * infinity turns a whole row into NaN, which the ELU's median instruction then drops), so the library reports it: every
 * forward pass tracks the largest first-layer pre-activation it saw, every (re)pack of an image checks the parameters, and a
 * violation sets a bit in the caller's device-side status word (mpg_cfg_t.status / mpg_wcache_t.status; nullable = not
 * reported).  A parameter beyond the bound enters the image clamped to +-65504 / 64.  The word is sticky: the caller reads
 * and clears it (mpg_amd.PolicyWithQs.check_status raises on it); `-DMPG_F32_MFMA` builds the exact-fp32 engine, which has
 * no envelope.  Reference counterpart: none (TensorFlow computes in float32 throughout). */
enum {
    MPG_STATUS_ACTIVATION_RANGE = 1, /* a first-hidden-layer activation reached 4094: that row's outputs are invalid */
    MPG_STATUS_PARAMETER_RANGE = 2,  /* a parameter reached 1023.5: clamped in the packed image */
    MPG_STATUS_NAN = 4               /* a NaN went into or came out of a network forward pass (mpg_policy_action and the other
                                        cfg-carrying forward entry points) - worker.py:95-107 `judge_is_nan` on the processed
                                        observations and on the actions, evaluated on the device */
};

/* Optional per-kernel timing with HIP events recorded on the launch stream.  mpg_prof_create allocates every event
 * up front (2 * MPG_PROF_SLOTS * max_samples of them), so that nothing is created inside a timed region;
 * mpg_prof_start(p, every) clears the samples and records around every `every`-th launch of each slot from then on
 * (0: stop; an event record is a stream packet of its own and costs ~4-5 us between two otherwise back-to-back
 * kernels, so timing EVERY launch slows a 0.5 ms step by 6 %); mpg_prof_read waits for the recorded events of a slot
 * and returns their summed duration and count.  A timer takes effect on the calls made with a mpg_cfg_t whose `prof`
 * field points to it (and on the native step driver's env launch).  Slots: 0 k_rollout_fwd, 1 k_rollout_bwd,
 * 2 env step, 3 k_forward, 4 k_backward, 5 k_wgrad, 6 k_target_fused, 7 k_critic_fused, 8 the caller's gradient exchange
 * (mpg_prof_region_begin / _end around whatever the caller enqueues between mpg_step_begin and mpg_step_end: RCCL all-reduce or
 * the one-shot exchange; the reference's hand-over is optimizer.py:60-94), 9 k_clip_adam_polyak.  No reference counterpart
 * (the reference times with utils/misc.py:39-90 TimerStat on the host).  Not thread-safe: one timer per launching
 * thread. */
enum { MPG_PROF_SLOTS = 10 };
typedef struct mpg_prof mpg_prof_t;
int mpg_prof_create(int max_samples, mpg_prof_t** out);
int mpg_prof_destroy(mpg_prof_t* p);
int mpg_prof_start(mpg_prof_t* p, int every);
int mpg_prof_read(mpg_prof_t* p, int slot, double* total_ms, int* count);
const char* mpg_prof_slot_name(int slot);
/* time a caller-side region on `stream` under `slot` with the same every-n-th sampling (no-ops for a null or stopped timer) */
int mpg_prof_region_begin(mpg_prof_t* p, int slot, mpg_stream_t stream);
int mpg_prof_region_end(mpg_prof_t* p, int slot, mpg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Vectorised real environment (K1)
 * ---------------------------------------------------------------------------------------------- */

/* Two real environments sit behind these entry points, selected by env_kind:
 *   MPG_ENV_PATH_TRACKING      PathTrackingEnv (path_tracking_env.py:356-487), act_dim 2, obs_dim = 6 + num_future_data
 *                              (0 <= num_future_data <= MPG_ENV_MAX_FUTURE): [v_x-20, v_y, r, dy, dphi, x | look-ahead
 *                              delta-y terms at x + k * 0.2 v_x, :385-402];
 *   MPG_ENV_INVERTED_PENDULUM  InvertedPendulumContiEnv (inverted_pendulum_conti.py:5-30 + inverted_pendulum_conti.xml) as
 *                              an analytic RK4 cart-pole (the reference steps it with MuJoCo, which cannot be pinned
 *                              here: csrc/env_cart_pole.hip), act_dim 1, obs_dim 4: [p, theta, pdot, thetadot].
 * state: the opaque block [MPG_ENV_STATE_DIM][n]. */
enum { MPG_ENV_MAX_FUTURE = 10 };

/* env.reset(init_obs=...)  - path_tracking_env.py:410-421 / DummyVecEnv.reset(init_obs=) for the pendulum.
 * Rebuilds the full state of all n agents from init_obs [n][obs_dim] (PathTracking reads its six base entries). */
int mpg_env_reset_from_obs(int env_kind, int n, int obs_dim, float* state, const float* init_obs, mpg_stream_t stream);

/* env.reset()  - path_tracking_env.py:423-454.  Re-draws every agent whose done_mask byte is
 * non-zero (done_mask == NULL: all agents) from the reset law x~U(0,600), dy~N(0,1), dphi~N(0,pi/9),
 * v_x~U(15,25), beta~N(0,.15), v_y=v_x tan(beta), r~N(0,.3) (pendulum: U(-0.01, 0.01) on all four,
 * inverted_pendulum_conti.py:21-25) with a counter-based Philox4x32-10 stream
 * keyed by (seed, call counter ctr, agent index); writes obs [n][obs_dim] for ALL agents. */
int mpg_env_reset(int env_kind, int n, int obs_dim, float* state, const uint8_t* done_mask, uint64_t seed, uint64_t ctr,
                  float* obs, mpg_stream_t stream);

/* PathTrackingEnv.step  - path_tracking_env.py:456-487 with VehicleDynamics.simulation :144-179 (20
 * sub-steps at 200 Hz), compute_rewards :181-199 (on the pre-step state), judge_done :474-487.
 * action [n][2] in [-1,1] (scaled by [1.2pi/9, 3] and clipped inside, :457-459); outputs obs [n][obs_dim],
 * reward [n], done [n] bytes.  done follows the reference literally and is therefore always 1
 * (SURVEY.md B-0); done_intended (nullable) receives the evidently intended |alpha| > |bound| test.
 * Pendulum: inverted_pendulum_conti.py:9-19, action [n][1] clipped to +-3, 2 RK4 steps of 0.02 s, real done flag. */
int mpg_env_step(int env_kind, int n, int obs_dim, float* state, const float* action, float* obs, float* reward,
                 uint8_t* done, uint8_t* done_intended, mpg_stream_t stream);

/* The inner body of OffPolicyWorker.sample after the policy (worker.py:108-112) in one launch: env.step, the
 * transition (obs, action, RAW reward, obs', done) written straight into the replay ring at (next_idx + i) % capacity
 * (ReplayBuffer.add, buffer.py:46-55), then env.reset() of the agents whose done flag is set
 * (path_tracking_env.py:445) with the Philox stream (seed, ctr).  Same results as mpg_env_step + mpg_replay_add +
 * mpg_env_reset.  obs_out [n][obs_dim]: the observations after the reset; done_out (nullable) [n]. */
int mpg_env_step_store_reset(int env_kind, int n, int obs_dim, float* state, const float* action, int capacity, int next_idx,
                             float* ring_obs, float* ring_act, float* ring_rew, float* ring_obs2, uint8_t* ring_done,
                             uint64_t seed, uint64_t ctr, float* obs_out, uint8_t* done_out, mpg_stream_t stream);

/* out[i] = ((slots[0][i] + slots[1][i]) + slots[2][i]) + ...  - the local sum of the one-shot all-reduce (SURVEY.md 8 f4;
 * mpg_amd/dist.py OneShotAllReduce): every rank of a node holds the ranks' gradient buffers in `n_slots` staging slots of
 * `n` floats and adds them in rank order, so that all replicas compute the same association.  Replaces the arithmetic of
 * the gradient exchange, optimizer.py:60-94 (there: one learner's gradients applied at a time). */
int mpg_sum_slots(const float* slots, int n_slots, int n, float* out, mpg_stream_t stream);
/* The same sum over a SLICE of the slots: slot r starts at slots + r * slot_stride (floats), n <= slot_stride entries are summed.
 * The reduce half of the two-shot (reduce-scatter + all-gather) form of the exchange: rank r sums only its 1/world slice, in the
 * same rank order, and distributes the result (mpg_amd/dist.py OneShotAllReduce, mode 'twoshot'; optimizer.py:60-94). */
int mpg_sum_slots_strided(const float* slots, int n_slots, size_t slot_stride, int n, float* out, mpg_stream_t stream);
/* The same sum with the clip's partial sums of squares of the RESULT as a by-product (round 6, ABI 10): sq_part[k*MPG_CLIP_PARTS + b]
 * exactly as mpg_sq_partials(out, seg_sizes, n_seg, ...) would compute them - the same bits - so that an exchanged gradient
 * (optimizer.py:60-94) reaches mpg_clip_adam_polyak / mpg_step_end (mpg_train_ctx_t.clip_partials_ready) without another pass over it.
 * seg_sizes[0..n_seg) (host): the networks at the head of the buffer; the n - sum(seg_sizes) floats behind them (statistics) are
 * summed too.  n_slots = 1: a copy with the partials (the gather array of the two-shot form). */
int mpg_sum_slots_sq(const float* slots, int n_slots, size_t slot_stride, int n, float* out, const int* seg_sizes, int n_seg,
                     float* sq_part /* n_seg * MPG_CLIP_PARTS floats */, mpg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Networks (K3), critic targets, losses and gradients (K5, K6)
 * ---------------------------------------------------------------------------------------------- */

/* Per-call options of the gradient entry points, reached through mpg_cfg_t.grad_opts (NULL: none).  HOST memory, read at
 * call time.  No reference counterpart: it only changes HOW the same numbers are scheduled. */
typedef struct {
    void* critics_ready_event;           /* hipEvent_t, nullable.  mpg_mpg_gradients then finishes the critics' gradient (chunk
                                            products + slab sums + loss sums) BEFORE the reverse sweep and records this event behind
                                            it on the launch stream: a caller that exchanges gradients between GPUs can start the
                                            critics' part of the exchange on a second stream under the reverse sweep
                                            (optimizer.py:60-94, payload mpg_learner.py:448-455).  Costs one launch more. */
} mpg_grad_opts_t;

/* Hyper-parameters of one config (host memory, read at call time). */
typedef struct {
    int obs_dim, act_dim;       /* path tracking: 6 + num_future_data (0 <= num_future_data <= MPG_ENV_MAX_FUTURE = 10), 2;
                                   pendulum: 4, 1.  LIMIT: the reference accepts any num_future_data
                                   (path_tracking_env.py:385-402); every entry point returns MPG_EINVAL for obs_dim > 16 (first
                                   layers: policy up to 16 wide, critics up to 24; obs_scale[16]) */
    int policy_out_act;         /* MPG_ACT_TANH (train_script.py:267) or MPG_ACT_LINEAR (train_script4mujoco.py:371) */
    float action_range;         /* <= 0: none; pendulum 3.0 -> a = range*tanh(mean) (policy.py:197-199) */
    float obs_scale[16];        /* 'scale' preprocessor, preprocessor.py:142-143 (train_script.py: [1,1,2,1,2.4,1/1200] + [1]*num_future_data) */
    float rew_scale, rew_shift; /* preprocessor.py:155-157 */
    float gamma;                /* 0.98 */
    int env_kind;               /* differentiable model used by the rollout: MPG_ENV_* */
    /* optional caller-owned handles (NULL: none) */
    const mpg_wcache_t* wcache[2]; /* packed images of up to two parameter vectors (e.g. networks, targets) that calls
                                      made with this cfg may receive pointers into */
    mpg_prof_t* prof;           /* kernel timer that the calls made with this cfg report to */
    int* status;                /* device, nullable: MPG_STATUS_* bits are OR-ed into it by the forward passes made with this cfg */
    const mpg_grad_opts_t* grad_opts; /* nullable: scheduling options of mpg_mpg_gradients (above) */
} mpg_cfg_t;

/* MLPNet.call  - model.py:39-43:  y[rows][out_used] = act(ELU(ELU(x W1 + b1) W2 + b2) W3 + b3)[:, :out_used].
 * x [rows][in_dim] (supported (in_dim, out_used): (6,2) (8,1) (4,1) (5,1) (6,1)); the first n_scaled input
 * columns are multiplied by in_scale[] (host array, may be NULL). */
int mpg_mlp_forward(const float* params, int in_dim, int out_dim, int out_used, int out_act, int rows,
                    const float* x, const float* in_scale, int n_scaled, float* y,
                    const mpg_wcache_t* wcache /* nullable */, mpg_stream_t stream);

/* PolicyWithQs.compute_action / compute_target_action, deterministic branch  - policy.py:193-217:
 * act[rows][act_dim] = mean half of the policy output on obs*obs_scale (x action_range*tanh if set).
 * explore_sigma > 0 adds N(0, sigma) per element from Philox(seed, ctr) - OffPolicyWorker.sample,
 * worker.py:96-98. */
int mpg_policy_action(const mpg_cfg_t* cfg, const float* policy_params, int rows, const float* obs,
                      float explore_sigma, uint64_t seed, uint64_t ctr, float* act, mpg_stream_t stream);

/* MPGLearner.compute_clipped_double_q_target  - learners/mpg_learner.py:126-134 (q2t != NULL), the 1-step
 * target of :148-152 (q2t == NULL), and TD3Learner.compute_clipped_double_q_target learners/td3.py:69-81 when
 * smooth_eps [rows][act_dim] (standard normal) is given: a' += clip(sigma*eps, -c, c) (no re-clipping).
 *   y = (rew+shift)*scale + gamma * min_i Qt_i(s~', a'),  a' = pi_t(s~').  No done mask. */
size_t mpg_q_targets_workspace_bytes(const mpg_cfg_t* cfg, int rows);
int mpg_q_targets(const mpg_cfg_t* cfg, const float* policy_t, const float* q1t, const float* q2t, int rows,
                  const float* rew, const float* obs_tp1, const float* smooth_eps, float smooth_sigma,
                  float smooth_clip, float* y, void* ws, size_t ws_bytes, mpg_stream_t stream);

/* TD3Learner.get_batch_data with a prioritized buffer (learners/td3.py:94-101): the clipped double-Q target y (td3.py:69-81,
 * as mpg_q_targets with smooth_eps) AND the plain Q1 target y1 of the priorities' td error (td3.py:83-92: y1 = r~ + gamma
 * Q1t(s~', pi_t(s~'))) from ONE evaluation of the target policy.  Same values as the two mpg_q_targets calls.  Workspace:
 * mpg_q_targets_workspace_bytes. */
int mpg_td3_targets(const mpg_cfg_t* cfg, const float* policy_t, const float* q1t, const float* q2t, int rows,
                    const float* rew, const float* obs_tp1, const float* smooth_eps, float smooth_sigma,
                    float smooth_clip, float* y, float* y1, void* ws, size_t ws_bytes, mpg_stream_t stream);

/* out[0 .. n) ~ N(0, 1) from Philox4x32-10(seed, ctr) + Box-Muller: the target-policy smoothing noise of TD3 (td3.py:74,
 * tf.random.normal in the reference) without a framework kernel; the same (seed, ctr) gives the same numbers in the native step
 * driver and in the method-by-method path. */
int mpg_normal_fill(int n, uint64_t seed, uint64_t ctr, float* out, mpg_stream_t stream);

/* TD3Learner.compute_td_error (td3.py:83-92) finished from what the step already has: out = (y1 - y) - td with td = Q1(s~, a) - y
 * from mpg_q_loss_grad and (y, y1) from mpg_td3_targets; signed (mpg_per_update takes |.| + eps). */
int mpg_td3_priority_errors(int rows, const float* y1, const float* y, const float* td, float* out, mpg_stream_t stream);

/* n-step return of MPG-v1 (mpg_learner.py:155-169) once the real-env rollout exists (mpg_env_step x n):
 *   y = sum_t gamma^t r~_t + gamma^n Q1t(s~_n, pi_t(s~_n)).  rewards [n][rows] RAW, last_obs [rows][obs_dim].
 * Workspace: mpg_q_targets_workspace_bytes. */
int mpg_nstep_targets(const mpg_cfg_t* cfg, const float* policy_t, const float* q1t, int rows, int n,
                      const float* rewards, const float* last_obs, float* y, void* ws, size_t ws_bytes,
                      mpg_stream_t stream);

/* MPGLearner.q_forward_and_backward  - mpg_learner.py:326-354 (also td3.py:103-118, nadp.py:173-184), ONE critic:
 *   L = 0.5 * mean_B (Q(s~,a) - y)^2 and dL/dtheta_Q.
 * inv_b_global = 1/B_global (the mean's divisor; 1/rows on one GPU).  Outputs: loss_sum[0] = this GPU's share
 * of L, grad = flat gradient (68353 floats path tracking) reduced over this GPU's rows and NOT clipped,
 * td (nullable) [rows] = Q(s~,a) - y. */
size_t mpg_q_loss_grad_workspace_bytes(const mpg_cfg_t* cfg, int rows);
int mpg_q_loss_grad(const mpg_cfg_t* cfg, const float* q_params, int rows, const float* obs, const float* act,
                    const float* y, float inv_b_global, float* loss_sum, float* grad, float* td, void* ws,
                    size_t ws_bytes, mpg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * n-step model rollout + mixed policy gradient (K2+K3+K4)
 * ---------------------------------------------------------------------------------------------- */

/* MPGLearner.model_rollout_for_policy_update + policy_forward_and_backward  - learners/mpg_learner.py:226-286,
 * :356-365 (default flags: deriv_interval_policy=False - parameter gradients flow only through the step-0
 * action) and, with all_steps_param_grad != 0, the NADP variant learners/nadp.py:128-171,
 * :186-194 where every step goes through pi_theta (needs rows*M % 16 == 0).
 *
 *   rows   B_local start states obs0 [rows][obs_dim]  (the M copies are tiled inside: trajectory = m*rows + b)
 *   eps    [n][M*rows] standard-normal draws for the model's injected noise
 *          (path_tracking_env.py:119: dy += 0.5 + 0.01 eps; inverted_pendulum_model.py:61: p += 0.1 + 0.5 eps);
 *          NULL: drawn inside the kernel from Philox4x32-10 keyed by (noise_seed, noise_ctr, step, trajectory)
 *   select/n_select/w (HOST arrays, n_select <= 4)   slices k whose mean returns R_k = mean(G_k + gamma^k Q1)
 *          enter the loss  sum_k w_k * (-R_k)   (w = rule_based_weights, mpg_learner.py:384-399, computed by the host)
 *   inv_b_global   1/B_global: divisor of the batch mean (the M-mean is applied inside); n < 32
 * Outputs
 *   ret_sum [n_select], ret_sqsum [n_select]: sum over this GPU's rows of the M-mean return and of its square
 *   grad    flat policy gradient (68612 floats path tracking) of the loss, reduced over this GPU's rows, NOT clipped.
 * Look-ahead observations (cfg->obs_dim > 6, num_future_data of path_tracking_env.py:246-270): the start observation's
 * look-ahead entries are inputs, every model observation's are copies of delta_y (:262-268) and their adjoint folds into
 * delta_y's; any M and either parameter-gradient mode (round 4). */
size_t mpg_rollout_pg_workspace_bytes(const mpg_cfg_t* cfg, int rows, int M, int n, int n_select,
                                      int all_steps_param_grad);
int mpg_rollout_pg(const mpg_cfg_t* cfg, const float* policy_params, const float* q1_params, int rows, int M,
                   int n, const int* select, int n_select, const float* w, const float* obs0, const float* eps,
                   uint64_t noise_seed, uint64_t noise_ctr, float inv_b_global, int all_steps_param_grad, float* ret_sum, float* ret_sqsum, float* grad,
                   void* ws, size_t ws_bytes, mpg_stream_t stream);

/* MPGLearner.compute_gradient without the clip, as one entry point  - learners/mpg_learner.py:401-431:
 * target (when y_in == NULL: clipped double-Q target from target_params, :126-134), critic losses and gradients of
 * the n_q (1: MPG-v1, 2: MPG-v2) critics (:326-354), n-step model rollout + mixed policy gradient (:226-286,356-365).
 * params / target_params / grad: flat [Q1 | (Q2) | policy] like PolicyWithQs.models (policy.py:72-86); y_in: optional
 * precomputed targets [rows] (MPG-v1's real-env n-step return); y_out [rows] receives the targets used.
 * stats (16 floats): [0..n_q) critic losses, [2..2+n_select) return sums, [2+n_select..2+2 n_select) sums of squares.
 * Everything is this GPU's UN-clipped partial, scaled by inv_b_global = 1/B_global: all-reduce, then
 * mpg_clip_by_global_norm.  Same results as mpg_q_targets + mpg_q_loss_grad + mpg_rollout_pg (up to the association
 * of the slab sums) in 6 launches instead of 22 when rows % 16 == 0 and M == 1; otherwise it calls those.
 * sq_part (nullable, (n_q+1)*MPG_CLIP_PARTS floats): on a single GPU the last launch also leaves the clip's partial sums
 * of squares of `grad` there (what mpg_sq_partials would compute), ready for mpg_clip_adam_polyak. */
/* Optional fused minibatch draw: ReplayBuffer.sample (buffer.py:70-78) from the device ring, the same draw as
 * mpg_replay_sample_uniform(n_storage, rows, seed, ctr, ...).  With `draw` != NULL the obs / act / rew / obs_tp1
 * arguments of mpg_mpg_gradients are OUTPUTS: the minibatch is gathered inside the first launch (one launch less per
 * iteration) and left there, the indices in idx_out and the dones (as float) in done_out (both nullable). */
typedef struct {
    int n_storage;                    /* transitions currently stored */
    uint64_t seed, ctr;               /* Philox key / counter of this draw (ctr = ReplayBuffer.replay_times) */
    const float *ring_obs, *ring_act, *ring_rew, *ring_obs2;
    const uint8_t* ring_done;
    int* idx_out;
    float* done_out;
    /* pre_gathered != 0: mpg_env_step_store_reset_draw already gathered every drawn row whose ring slot lies OUTSIDE the
     * window [fresh_start, fresh_start + fresh_count) mod capacity (the slots that launch was writing) into the output
     * arrays; only rows drawn from the window are gathered now.  Same minibatch either way. */
    int pre_gathered, capacity, fresh_start, fresh_count;
} mpg_replay_draw_t;

/* mpg_env_step_store_reset plus, in spare workgroups of the same launch, the gather of the NEXT minibatch draw
 * (draw->n_storage = the ring size after this add, draw->seed / ctr = that draw's Philox key and counter): the random
 * ring reads (one DRAM line and one page-table walk per array and row) then overlap the 20 sub-steps of the env instead
 * of sitting at the head of mpg_mpg_gradients' first launch.  Rows drawn from the slots this launch writes are skipped;
 * pass the same draw with pre_gathered = 1, fresh_start = next_idx, fresh_count = n to mpg_mpg_gradients.
 * Path-tracking env with obs_dim 6 only (MPG_EINVAL otherwise). */
int mpg_env_step_store_reset_draw(int env_kind, int n, int obs_dim, float* state, const float* action, int capacity, int next_idx,
                                  float* ring_obs, float* ring_act, float* ring_rew, float* ring_obs2, uint8_t* ring_done,
                                  uint64_t seed, uint64_t ctr, float* obs_out, uint8_t* done_out /* nullable */,
                                  const mpg_replay_draw_t* draw, int rows, float* b_obs, float* b_act, float* b_rew,
                                  float* b_obs2, mpg_stream_t stream);

/* OffPolicyWorker.sample's inner body (worker.py:95-112) as ONE launch for the path-tracking env with six-entry observations:
 * mpg_policy_action (deterministic action + N(0, explore_sigma) noise keyed by noise_seed / noise_ctr, written to act_out [n][2])
 * followed by mpg_env_step_store_reset (draw == NULL) or mpg_env_step_store_reset_draw - the policy pass of a 16-agent group by one
 * workgroup, whose first wave then steps those agents.  obs_io [n][6]: the current observations in, the next ones out.  Actions,
 * ring rows, env state and observations are bit-identical to the two stand-alone calls'.  MPG_EINVAL for any other env / width. */
int mpg_worker_step(const mpg_cfg_t* cfg, const float* policy_params, int n, float* state, float* obs_io, float explore_sigma,
                    uint64_t noise_seed, uint64_t noise_ctr, float* act_out, int capacity, int next_idx, float* ring_obs,
                    float* ring_act, float* ring_rew, float* ring_obs2, uint8_t* ring_done, uint64_t env_seed, uint64_t env_ctr,
                    uint8_t* done_out /* nullable */, const mpg_replay_draw_t* draw /* nullable */, int rows, float* b_obs,
                    float* b_act, float* b_rew, float* b_obs2, mpg_stream_t stream);

size_t mpg_mpg_gradients_workspace_bytes(const mpg_cfg_t* cfg, int rows, int M, int n, int n_select, int n_q);
int mpg_mpg_gradients(const mpg_cfg_t* cfg, int n_q, const float* params, const float* target_params, int rows,
                      float* obs, float* act, float* rew, float* obs_tp1, const float* y_in,
                      int M, int n, const int* select, int n_select, const float* w, const float* eps,
                      uint64_t noise_seed, uint64_t noise_ctr, float inv_b_global, float* grad, float* stats,
                      float* y_out, float* sq_part /* nullable: see mpg_sq_partials */,
                      const mpg_replay_draw_t* draw /* nullable */, void* ws, size_t ws_bytes, mpg_stream_t stream);

/* NADPLearner.model_rollout_for_q_estimation  - learners/nadp.py:87-126: from (s, a_replay) roll n model steps,
 * later actions from pi_theta, y = G_n + gamma^n * Q1_target(s~_n, pi_theta(s~_n)) (no gradient).
 * eps [n][rows] standard normal, or NULL for in-kernel Philox(noise_seed, noise_ctr) draws. */
size_t mpg_rollout_q_target_workspace_bytes(const mpg_cfg_t* cfg, int rows);
int mpg_rollout_q_target(const mpg_cfg_t* cfg, const float* policy_params, const float* q1t, int rows, int n,
                         const float* obs0, const float* act0, const float* eps, uint64_t noise_seed,
                         uint64_t noise_ctr, float* y, void* ws, size_t ws_bytes, mpg_stream_t stream);

/* MPGLearner.model_rollout_for_q_estimation  - learners/mpg_learner.py:180-224 (the heuristic-bias rollout): tile
 * (s, a_replay) M times, roll the model max(select) steps - first action a_replay, later ones from pi_theta - and
 * bootstrap every selected slice k with gamma^k * Q1_target(s~_k, a_k); the InvertedPendulum branch clips that Q to
 * [-0.5, 0] for every slice but the first (:206-209).  y [n_select][rows]: mean over the M copies, slices in the order of
 * `select` (HOST array, n_select <= 4) - the reference's concatenation.  eps [max(select)][M*rows] standard normal, or
 * NULL for in-kernel Philox(noise_seed, noise_ctr) draws.  No gradient. */
size_t mpg_rollout_q_estimation_workspace_bytes(const mpg_cfg_t* cfg, int rows, int M, int n_select);
int mpg_rollout_q_estimation(const mpg_cfg_t* cfg, const float* policy_params, const float* q1t, int rows, int M,
                             const int* select, int n_select, const float* obs0, const float* act0, const float* eps,
                             uint64_t noise_seed, uint64_t noise_ctr, float* y, void* ws, size_t ws_bytes,
                             mpg_stream_t stream);

/* TD3Learner.policy_forward_and_backward  - learners/td3.py:120-134:
 *   loss = -mean_B min(Q1,Q2)(s~, pi(s~)); grad = flat policy gradient (unclipped, reduced over this GPU's rows,
 *   divisor B_global through inv_b_global); qmin_sum / qmin_sqsum: sum and sum of squares of min-Q over the rows
 *   (value_mean / value_var, td3.py:130-132).  The gradient flows through whichever critic is smaller per row. */
size_t mpg_td3_policy_grad_workspace_bytes(const mpg_cfg_t* cfg, int rows);
int mpg_td3_policy_grad(const mpg_cfg_t* cfg, const float* policy_params, const float* q1, const float* q2,
                        int rows, const float* obs, float inv_b_global, float* qmin_sum, float* qmin_sqsum,
                        float* grad, void* ws, size_t ws_bytes, mpg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * clip_by_global_norm + Keras Adam + Polyak (K7, K8) over the flat [net0 | net1 | ...] vectors
 * ---------------------------------------------------------------------------------------------- */

/* tf.clip_by_global_norm(g, clip) applied to each of the n_seg (<= 8) consecutive networks of `grad`
 * (seg_sizes: HOST array of their lengths) - learners/mpg_learner.py:415-431: norms[k] = ||g_k||_2,
 * g_k *= clip * min(1/norm, 1/clip) in place.  nonfinite_flags (device int[n_seg], nullable) receives 1 for
 * every network whose norm is not finite, else 0 (optimizer.py:357-361 zeroes such gradient lists).  Must run
 * AFTER the cross-GPU all-reduce: the clip is not linear. */
#define MPG_CLIP_PARTS 272 /* partial sums of squares per network: one per 256 elements up to 69 632, strided beyond */
int mpg_clip_by_global_norm(float* grad, const int* seg_sizes, int n_seg, float clip, float* norms,
                            int* nonfinite_flags,
                            float* scratch /* nullable: n_seg*MPG_CLIP_PARTS floats -> parallel two-launch form */,
                            mpg_stream_t stream);

/* The first half of the parallel clip on its own: sq_part[k*MPG_CLIP_PARTS + b] = sum of squares of block b
 * (elements 256 b ... 256 b + 255, + multiples of 256*MPG_CLIP_PARTS) of network k, fixed summation order. */
int mpg_sq_partials(const float* grad, const int* seg_sizes, int n_seg, float* sq_part, mpg_stream_t stream);

/* mpg_clip_by_global_norm's second half + mpg_adam_polyak in ONE launch: norms from sq_part (mpg_sq_partials or
 * mpg_mpg_gradients' by-product), grad scaled in place (it is the list compute_gradient returns), NaN guard
 * (optimizer.py:357-361: any non-finite norm -> every Adam step sees zeros), Adam, Polyak.  Bit-identical to
 * mpg_clip_by_global_norm(scratch) followed by mpg_adam_polyak(skip_flags = nonfinite_flags). */
int mpg_clip_adam_polyak(float* w, float* m, float* v, float* target, float* grad, const float* sq_part,
                         const int* seg_sizes, int n_seg, float clip, const float* lr_t, const int* do_adam,
                         const int* do_polyak, float tau, float* norms, int* nonfinite_flags,
                         const mpg_wcache_t* wc_w /* nullable: packed images of w, kept current */,
                         const mpg_wcache_t* wc_target /* nullable: same for target */, mpg_stream_t stream);

/* PolicyWithQs.apply_gradients + update_*_target  - policy.py:123-171.  For every network k (HOST arrays):
 * do_adam[k]: one Keras Adam step (beta .9/.999, eps 1e-7 outside the sqrt, TF ApplyAdam form) with the
 * bias-corrected rate lr_t[k] = lr(step)*sqrt(1-b2^t)/(1-b1^t) computed by the caller from ITS per-optimizer
 * step counter and PolynomialDecay schedule (policy.py:54-70); do_polyak[k]: target = tau*w + (1-tau)*target
 * afterwards.  If any of skip_flags[0..n_skip_flags) (device ints, nullable) is non-zero the gradient is taken as
 * zeros (NaN guard, optimizer.py:357-361). */
int mpg_adam_polyak(float* w, float* m, float* v, float* target, const float* grad, const int* seg_sizes,
                    int n_seg, const float* lr_t, const int* do_adam, const int* do_polyak, float tau,
                    const int* skip_flags, int n_skip_flags,
                    const mpg_wcache_t* wc_w /* nullable: packed images of w, kept current */,
                    const mpg_wcache_t* wc_target /* nullable: same for target */, mpg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * On-device replay ring (K9, uniform part)  - buffer.py:21-91
 * ---------------------------------------------------------------------------------------------- */

/* ReplayBuffer.add_batch (buffer.py:46-55,80-82): writes n transitions at ring positions
 * (next_idx + i) % capacity.  Ring arrays: obs [cap][obs_dim], act [cap][act_dim], rew [cap],
 * obs2 [cap][obs_dim], done [cap] bytes - RAW observations and rewards. */
int mpg_replay_add(int capacity, int next_idx, int n, int obs_dim, int act_dim, const float* s_obs,
                   const float* s_act, const float* s_rew, const float* s_obs2, const uint8_t* s_done,
                   float* obs, float* act, float* rew, float* obs2, uint8_t* done, mpg_stream_t stream);

/* ReplayBuffer._encode_sample (buffer.py:57-68): gathers rows idx[i] of the ring; o_done (nullable) is
 * written as float32 like the learners' cast (mpg_learner.py:71). */
int mpg_replay_gather(int n, const int* idx, int obs_dim, int act_dim, const float* obs, const float* act,
                      const float* rew, const float* obs2, const uint8_t* done, float* o_obs, float* o_act,
                      float* o_rew, float* o_obs2, float* o_done, mpg_stream_t stream);

/* ReplayBuffer.sample (buffer.py:70-78) in one launch: mpg_uniform_indices + mpg_replay_gather (same Philox stream,
 * same results). */
int mpg_replay_sample_uniform(int n_storage, int n, uint64_t seed, uint64_t ctr, int obs_dim, int act_dim,
                              const float* obs, const float* act, const float* rew, const float* obs2,
                              const uint8_t* done, int* idx, float* o_obs, float* o_act, float* o_rew, float* o_obs2,
                              float* o_done, mpg_stream_t stream);

/* ReplayBuffer.sample_idxes (buffer.py:70-71): n indices uniform in [0, n_storage), with replacement,
 * from Philox(seed, ctr). */
int mpg_uniform_indices(int n_storage, int n, uint64_t seed, uint64_t ctr, int* idx, mpg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * On-device prioritized replay (K9, proportional part)  - buffer.py:94-189, utils/segment_tree.py:13-151
 * ---------------------------------------------------------------------------------------------- */

/* Each tree is ONE float64 array of 2*capacity nodes in the reference's heap order (node 1 = root, leaves at
 * [capacity, 2*capacity), capacity a power of two, segment_tree.py:40-44); float64 because the reference keeps
 * python floats.  stamp: int scratch [capacity] owned by the caller (duplicate-index resolution). */

/* SegmentTree.__init__: neutral elements 0.0 / +inf (segment_tree.py:106,146); stamp = -1. */
int mpg_per_init(double* sum_tree, double* min_tree, int* stamp, int capacity, mpg_stream_t stream);

/* Batched SegmentTree.__setitem__ (segment_tree.py:90-97) = PrioritizedReplayBuffer.update_priorities / add
 * (buffer.py:127-136,166-189): leaf[idx[i]] = (|prio[i]| + eps)^alpha in both trees (the reference asserts
 * priority > 0; learners hand signed td errors, SURVEY.md B-3, hence |.| + eps), then every internal node is
 * recomputed as left + right / min(left, right).  Duplicates: the last entry of the batch wins, like the
 * reference's sequential loop.  max_priority (device DOUBLE since ABI 10 - the reference's _max_priority is a python float -,
 * nullable) tracks max(|prio| + eps) (buffer.py:189). */
int mpg_per_update(double* sum_tree, double* min_tree, int* stamp, int capacity, int n, const int* idx,
                   const float* prio, double alpha, double eps, double* max_priority, mpg_stream_t stream);

/* PrioritizedReplayBuffer.add for n consecutive ring slots start, start + 1, ... (mod ring_capacity): their leaves enter at the
 * current max priority, (*max_priority) ** alpha in float64 (buffer.py:127-136).  idx_scratch: n ints of device scratch. */
int mpg_per_add(double* sum_tree, double* min_tree, int* stamp, int capacity, int ring_capacity, int start, int n,
                double alpha, double* max_priority, int* idx_scratch, mpg_stream_t stream);

/* PrioritizedReplayBuffer._sample_proportional + IS weights (buffer.py:138-160):
 *   mass_i = u_i * sum(0, n_storage)  (inclusive end, buffer.py:141);  idx_i = find_prefixsum_idx(mass_i)
 *   (segment_tree.py:114-140);  w_i = (p_i/sum * n_storage)^-beta / max_w  with max_w from the min tree.
 * u [n] uniform(0,1) float64 draws supplied by the caller (NULL: 53-bit Philox(seed, ctr) draws). */
int mpg_per_sample(const double* sum_tree, const double* min_tree, int capacity, int n_storage, int n,
                   const double* u, uint64_t seed, uint64_t ctr, double beta, int* idx, float* is_weight,
                   mpg_stream_t stream);
/* The same draw with the gather of the sampled transitions in the same launch: PrioritizedReplayBuffer.sample, buffer.py:161-164
 * (_sample_proportional + IS weights + _encode_sample).  Results identical to mpg_per_sample followed by mpg_replay_gather.
 * o_done (nullable): the dones as float, like mpg_replay_gather. */
int mpg_per_sample_gather(const double* sum_tree, const double* min_tree, int capacity, int n_storage, int n,
                          const double* u, uint64_t seed, uint64_t ctr, double beta, int* idx, float* is_weight,
                          int obs_dim, int act_dim, const float* ring_obs, const float* ring_act, const float* ring_rew,
                          const float* ring_obs2, const uint8_t* ring_done, float* o_obs, float* o_act, float* o_rew,
                          float* o_obs2, float* o_done, mpg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Native step driver  - SingleProcessOffPolicyOptimizer.step, optimizer.py:330-362
 * ---------------------------------------------------------------------------------------------- */

/* One training iteration of the MPG learner (round 4: also NADP and TD3 + prioritized replay) enqueued by native code, so that the ~45 kernel launches of a step are
 * not paced by the Python interpreter.  It calls exactly the entry points above, in the reference's order:
 *   mpg_step_begin:  [every sampling_interval-th iteration: sample_iters x (policy + N(0,sigma) -> env.step ->
 *                    ring add -> env.reset)]  (worker.py:91-119, optimizer.py:332-337);  replay (uniform indices +
 *                    gather, buffer.py:70-91);  target (MPG-v2: mpg_learner.py:126-134; MPG-v1: 25 real-env steps +
 *                    n-step return, :146-169);  Q loss/gradients (:326-354);  model rollout + mixed policy gradient
 *                    (:226-286,356-365).  Leaves UN-clipped partials scaled by 1/(batch*world_size) in grad[].
 *   (the caller all-reduces grad[0 .. n_grad+16) across GPUs here)
 *   mpg_step_end:    per-network clip_by_global_norm (:415-431) + Adam/Polyak (policy.py:123-171).
 * All pointers are device buffers owned by the caller; counters are advanced in place (host fields). */
typedef struct {
    mpg_cfg_t cfg;
    int learner_version;              /* 1: MPG-v1 (n-step real-env target), 2: MPG-v2 (clipped double-Q target),
                                         3: NADP (learners/nadp.py:209-241: networks [Q1 | policy]; n = the Q-target AND policy
                                            rollout horizon, n_select / select / eta / total_ite unused),
                                         4: TD3 (learners/td3.py:150-188: networks [Q1 | Q2 | policy]; n, M, select unused) */
    int num_agent, sample_iters;      /* worker: sample_iters env steps of num_agent agents per sampling call */
    int sampling_interval;            /* optimizer.py:331 (10 in the reference) */
    int batch, n, M, n_select, select[4];
    float eta;                        /* rule_based_weights, mpg_learner.py:384-399 */
    int total_ite;
    float clip, tau;
    int delay_update, num_batch_reuse, world_size;
    int grads_exchanged;              /* != 0: the caller sums grad[] across processes between mpg_step_begin and mpg_step_end
                                         (every run with world_size > 1): the clip's partial sums of squares are then taken
                                         from the exchanged gradient in mpg_step_end instead of being a by-product of
                                         mpg_step_begin's last launch.  Same partials, same order, either way. */
    float explore_sigma;
    float value_lr[3], policy_lr[3];  /* PolynomialDecay(lr0, steps, lr_end), policy.py:54-70 */
    /* counters */
    uint64_t worker_seed, noise_ctr, env_seed, env_ctr, replay_seed, replay_times, learner_seed, learner_counter;
    int ring_capacity, ring_next, ring_size;
    long long opt_steps[3];           /* per-optimizer step counters in network order Q1,(Q2),policy */
    /* worker / env */
    float *env_state, *w_obs, *w_act, *w_rew, *w_obs2;
    uint8_t *w_done, *w_done_intended;
    /* replay ring and the sampled minibatch */
    float *ring_obs, *ring_act, *ring_rew, *ring_obs2;
    uint8_t* ring_done;
    int* idx;
    float *b_obs, *b_act, *b_rew, *b_obs2, *b_done, *b_targets;
    /* networks [Q1 | (Q2) | policy], their targets, Adam moments */
    float *params, *targets, *adam_m, *adam_v;
    float* grad;                      /* n_grad + 16 floats: gradients then statistics (q losses, return sums) */
    float* norms;                     /* n_nets */
    float* clip_scratch;              /* n_nets * MPG_CLIP_PARTS floats */
    int* nonfinite;                   /* n_nets */
    /* MPG-v1 only: the learner's own env for the n-step sampler (batch agents) */
    float *l_env_state, *l_obs, *l_act, *l_rewards;
    uint8_t *l_done, *l_done_intended;
    void *ws0, *ws1;
    size_t ws0_bytes, ws1_bytes;
    /* TD3 (learner_version 4) */
    float smooth_sigma, smooth_clip;  /* target policy smoothing, td3.py:74-76 (noise: mpg_normal_fill(learner_seed, learner_counter)) */
    int prioritized;                  /* != 0: PrioritizedReplayBuffer (buffer.py:94-189) - the fields below; 0: uniform replay */
    double *per_sum, *per_min;        /* segment trees, 2 * per_capacity doubles each (mpg_per_init) */
    int* per_stamp;
    int per_capacity;
    double* per_max_priority;         /* float64 (ABI 10): the reference's _max_priority is a python float */
    double per_alpha, per_beta, per_eps;
    float* b_weights;                 /* [batch] IS weights of the draw (buffer.py:146-160; the TD3 loss does not use them) */
    float* scratch;                   /* max(batch * (act_dim + 3), 2 * num_agent) floats: smoothing noise | y1 | td | priority errors
                                         (and the index / priority pairs of a ring add) */
    /* scheduling option (round 5, ABI 9; MPG only, optional) */
    void* critics_ready_event;        /* hipEvent_t, nullable: see mpg_grad_opts_t (the caller overlaps the critics' exchange) */
    mpg_grad_opts_t grad_opts;        /* storage for cfg.grad_opts during mpg_step_begin */
    /* round 6, ABI 10 */
    int clip_partials_ready;          /* != 0 (meaningful with grads_exchanged): the caller's exchange left the clip partials of the
                                         REDUCED gradient in clip_scratch (mpg_sum_slots_sq) - mpg_step_end takes them as given
                                         instead of re-reading grad[] (mpg_sq_partials).  `grad` may differ between mpg_step_begin
                                         (where the partial gradient is written: e.g. the caller's own staging slot of the
                                         exchange) and mpg_step_end (the reduced buffer): it is read at call time. */
} mpg_train_ctx_t;

/* workspace requirements of a context (ws0: targets / critic; ws1: rollout) */
int mpg_step_workspace_bytes(const mpg_train_ctx_t* ctx, size_t* ws0_bytes, size_t* ws1_bytes);
int mpg_step_begin(mpg_train_ctx_t* ctx, int iteration, mpg_stream_t stream);
int mpg_step_end(mpg_train_ctx_t* ctx, int iteration, mpg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MPG_HIP_H */
