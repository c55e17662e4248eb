// Native step driver: SingleProcessOffPolicyOptimizer.step (optimizer.py:330-362) for the MPG learner, built purely
// from the public entry points of this library so that Python is not on the launch path.
#include <algorithm>
#include <cmath>

#include "mpg_common.h"

namespace {

constexpr int H = MPG_HIDDEN;
inline int net_size(int in_dim, int out_dim) { return in_dim * H + H + H * H + H + H * out_dim + out_dim; }

struct Layout {
    int n_nets, q_size, p_size, n_grad;
    int sizes[3], off[3];   // Q1, (Q2), policy
};

Layout layout(const mpg_train_ctx_t* c) {
    Layout l;
    l.q_size = net_size(c->cfg.obs_dim + c->cfg.act_dim, 1);
    l.p_size = net_size(c->cfg.obs_dim, 2 * c->cfg.act_dim);
    l.n_nets = (c->learner_version == 2 || c->learner_version == 4) ? 3 : 2;        // MPG-v2 / TD3: double Q
    int o = 0;
    for (int k = 0; k < l.n_nets; ++k) {
        l.sizes[k] = k == l.n_nets - 1 ? l.p_size : l.q_size;
        l.off[k] = o;
        o += l.sizes[k];
    }
    l.n_grad = o;
    return l;
}

// MPGLearner.rule_based_weights, mpg_learner.py:384-399, in float32 like the TF graph
void rule_based_weights(int ite, int total_ite, float eta, const int* select, int ns, float* w) {
    float lam = (1.f - eta) + (2.f * eta / (float)total_ite) * (float)ite;
    lam = std::min(std::max(lam, 0.f), 1.5f);
    int mx = 0;
    for (int k = 0; k < ns; ++k) mx = std::max(mx, select[k]);
    float inv[4], m = -INFINITY;
    for (int k = 0; k < ns; ++k) {
        const float bias = lam < 1.f ? powf(lam, (float)select[k]) : powf(2.f - lam, (float)(mx - select[k]));
        inv[k] = 1.f / (bias + 1e-8f);
        m = std::max(m, inv[k]);
    }
    float sum = 0.f;
    for (int k = 0; k < ns; ++k) { w[k] = expf(inv[k] - m); sum += w[k]; }
    for (int k = 0; k < ns; ++k) w[k] /= sum;
}

// PolynomialDecay (policy.py:54,62) and the ApplyAdam step size AS TENSORFLOW FORMS THEM: every operand a float32 tensor
// (schedule in the dtype of the initial rate; beta^t = pow of the float32 hyper-parameter; alpha = lr sqrt(1 - b2^t) / (1 - b1^t)
// in float32).  float32(0.999) is 1.3e-8 above 0.999, which puts alpha 6.7e-6 below the real-number formula for the first
// thousands of steps - found in round 6 when the reference's own PolicyWithQs.apply_gradients first ran against this path.
// beta^t: double-precision pow of the float32 operand rounded once = the correctly rounded powf.
float polynomial_decay(const float* sched, long long step) {
    const float lr0 = sched[0], S = sched[1], lr_end = sched[2];
    const float p = std::min((float)step, S) / S;
    return (lr0 - lr_end) * (1.f - p) + lr_end;
}

float adam_step_size(const float* sched, long long steps_done) {
    const float lr = polynomial_decay(sched, steps_done);
    const double t = (double)(steps_done + 1);
    const float b1p = (float)std::pow((double)0.9f, t), b2p = (float)std::pow((double)0.999f, t);
    return lr * std::sqrt(1.f - b2p) / (1.f - b1p);
}

#define TRY(call)            \
    do {                     \
        int rc_ = (call);    \
        if (rc_) return rc_; \
    } while (0)

inline bool exchanged(const mpg_train_ctx_t* c) { return c->world_size > 1 || c->grads_exchanged != 0; }

inline bool is_mpg(const mpg_train_ctx_t* c) { return c->learner_version == 1 || c->learner_version == 2; }

bool ctx_ok(const mpg_train_ctx_t* c) {
    if (!c || c->learner_version < 1 || c->learner_version > 4) return false;
    const bool common = c->num_agent > 0 && c->batch > 0 && c->world_size > 0 && c->sampling_interval > 0 && c->num_batch_reuse > 0 &&
                        c->ring_capacity > 0 && c->params && c->targets && c->grad && c->ws0 && c->ws1;
    if (!common) return false;
    if (is_mpg(c))
        return c->n > 0 && c->M > 0 && c->n_select > 0 && c->n_select <= 4 && (c->learner_version == 2 ? 2 : 1) + 2 * c->n_select <= 8;
    if (c->learner_version == 3) return c->n > 0 && c->num_batch_reuse == 1;
    // TD3: the scratch block, and the trees when the replay is prioritized
    return c->scratch && c->num_batch_reuse == 1 &&
           (!c->prioritized || (c->per_sum && c->per_min && c->per_stamp && c->per_capacity >= c->ring_capacity && c->per_max_priority && c->b_weights));
}

// ---- worker.sample + replay_buffer.add_batch (optimizer.py:332-337, worker.py:91-119), the plain form (NADP, TD3) ----
// worker.py:95-112 as one launch (mpg_worker_step) where it exists: path-tracking env, six-entry observations
inline bool worker_step_fused(const mpg_train_ctx_t* c) {
    return c->cfg.env_kind == MPG_ENV_PATH_TRACKING && c->cfg.obs_dim == 6 && c->cfg.act_dim == 2;
}

int sample_and_add(mpg_train_ctx_t* c, const float* policy, mpg_stream_t s) {
    const int od = c->cfg.obs_dim, kind = c->cfg.env_kind;
    for (int it = 0; it < c->sample_iters; ++it) {
        MPG_REQUIRE(c->num_agent <= c->ring_capacity, "mpg_step_begin: ring smaller than one sample");
        if (worker_step_fused(c)) {     // policy pass + env.step -> ring -> env.reset in one launch
            TRY(mpg_worker_step(&c->cfg, policy, c->num_agent, c->env_state, c->w_obs, c->explore_sigma, c->worker_seed, c->noise_ctr++,
                                c->w_act, c->ring_capacity, c->ring_next, c->ring_obs, c->ring_act, c->ring_rew, c->ring_obs2, c->ring_done,
                                c->env_seed, c->env_ctr++, c->w_done, nullptr, 0, nullptr, nullptr, nullptr, nullptr, s));
        } else {
        TRY(mpg_policy_action(&c->cfg, policy, c->num_agent, c->w_obs, c->explore_sigma, c->worker_seed, c->noise_ctr++, c->w_act, s));
        mpg_prof_begin(c->cfg.prof, 2, mpg_stream(s));
        TRY(mpg_env_step_store_reset(kind, c->num_agent, od, c->env_state, c->w_act, c->ring_capacity, c->ring_next, c->ring_obs,
                                     c->ring_act, c->ring_rew, c->ring_obs2, c->ring_done, c->env_seed, c->env_ctr++, c->w_obs, c->w_done, s));
        mpg_prof_end(c->cfg.prof, 2, mpg_stream(s));
        }
        if (c->learner_version == 4 && c->prioritized)       // new transitions enter at the max priority (buffer.py:127-136)
            TRY(mpg_per_add(c->per_sum, c->per_min, c->per_stamp, c->per_capacity, c->ring_capacity, c->ring_next, c->num_agent,
                            c->per_alpha, c->per_max_priority, reinterpret_cast<int*>(c->scratch), s));
        c->ring_next = (c->ring_next + c->num_agent) % c->ring_capacity;
        c->ring_size = std::min(c->ring_size + c->num_agent, c->ring_capacity);
    }
    return MPG_OK;
}

// ---- NADPLearner.compute_gradient, learners/nadp.py:209-241 (networks [Q1 | policy]) ----
int nadp_gradients(mpg_train_ctx_t* c, const Layout& l, mpg_stream_t s) {
    const float* q1 = c->params;
    const float* policy = c->params + l.off[1];
    const float* q1t = c->targets;
    const float inv_b = 1.f / ((float)c->batch * (float)c->world_size);
    float* stats = c->grad + l.n_grad;
    // n-step model-rollout target from the stored (s, a) (nadp.py:87-126), critic loss (:173-184), policy loss -R_n with the
    // parameter gradient through all n + 1 evaluations (:128-194); noise counters as NADPLearner does (2 k, 2 k + 1)
    TRY(mpg_rollout_q_target(&c->cfg, policy, q1t, c->batch, c->n, c->b_obs, c->b_act, nullptr, c->learner_seed, 2 * c->learner_counter,
                             c->b_targets, c->ws0, c->ws0_bytes, s));
    TRY(mpg_q_loss_grad(&c->cfg, q1, c->batch, c->b_obs, c->b_act, c->b_targets, inv_b, stats, c->grad + l.off[0], nullptr, c->ws0,
                        c->ws0_bytes, s));
    const int sel[2] = {0, c->n};
    const float w[2] = {0.f, 1.f};
    return mpg_rollout_pg(&c->cfg, policy, q1, c->batch, 1, c->n, sel, 2, w, c->b_obs, nullptr, c->learner_seed, 2 * c->learner_counter + 1,
                          inv_b, 1, stats + 2, stats + 4, c->grad + l.off[1], c->ws1, c->ws1_bytes, s);
}

// ---- TD3Learner.compute_gradient, learners/td3.py:150-188 (networks [Q1 | Q2 | policy]) + the priority update of
//      optimizer.py:351-353 ----
int td3_gradients(mpg_train_ctx_t* c, const Layout& l, mpg_stream_t s) {
    const int ad = c->cfg.act_dim, B = c->batch;
    const float *q1 = c->params, *q2 = c->params + l.off[1], *policy = c->params + l.off[2];
    const float *q1t = c->targets, *q2t = c->targets + l.off[1], *policy_t = c->targets + l.off[2];
    const float inv_b = 1.f / ((float)B * (float)c->world_size);
    float* stats = c->grad + l.n_grad;
    float* eps = c->scratch;                          // [B][ad] smoothing noise
    float* y1 = eps + (size_t)B * ad;                 // [B] the priorities' plain Q1 target
    float* td = y1 + B;                               // [B] Q1(s, a) - y
    float* perr = td + B;                             // [B] y1 - Q1(s, a)
    TRY(mpg_normal_fill(B * ad, c->learner_seed, c->learner_counter, eps, s));
    TRY(mpg_td3_targets(&c->cfg, policy_t, q1t, q2t, B, c->b_rew, c->b_obs2, eps, c->smooth_sigma, c->smooth_clip, c->b_targets, y1,
                        c->ws0, c->ws0_bytes, s));
    TRY(mpg_q_loss_grad(&c->cfg, q1, B, c->b_obs, c->b_act, c->b_targets, inv_b, stats, c->grad + l.off[0], td, c->ws0, c->ws0_bytes, s));
    TRY(mpg_q_loss_grad(&c->cfg, q2, B, c->b_obs, c->b_act, c->b_targets, inv_b, stats + 1, c->grad + l.off[1], nullptr, c->ws0,
                        c->ws0_bytes, s));
    if (c->prioritized) {
        TRY(mpg_td3_priority_errors(B, y1, c->b_targets, td, perr, s));
        TRY(mpg_per_update(c->per_sum, c->per_min, c->per_stamp, c->per_capacity, B, c->idx, perr, c->per_alpha, c->per_eps,
                           c->per_max_priority, s));
    }
    return mpg_td3_policy_grad(&c->cfg, policy, q1, q2, B, c->b_obs, inv_b, stats + 2, stats + 3, c->grad + l.off[2], c->ws1, c->ws1_bytes, s);
}

}  // namespace

extern "C" int mpg_step_workspace_bytes(const mpg_train_ctx_t* c, size_t* ws0, size_t* ws1) {
    MPG_REQUIRE(c && ws0 && ws1, "mpg_step_workspace_bytes: null pointer");
    *ws0 = std::max(mpg_q_targets_workspace_bytes(&c->cfg, c->batch), mpg_q_loss_grad_workspace_bytes(&c->cfg, c->batch));
    if (c->learner_version == 3) {
        *ws0 = std::max(*ws0, mpg_rollout_q_target_workspace_bytes(&c->cfg, c->batch));
        *ws1 = mpg_rollout_pg_workspace_bytes(&c->cfg, c->batch, 1, c->n, 2, 1);
    } else if (c->learner_version == 4) {
        *ws1 = mpg_td3_policy_grad_workspace_bytes(&c->cfg, c->batch);
    } else
    *ws1 = mpg_mpg_gradients_workspace_bytes(&c->cfg, c->batch, c->M, c->n, c->n_select, c->learner_version == 2 ? 2 : 1);
    MPG_REQUIRE(*ws0 && *ws1, "mpg_step_workspace_bytes: unsupported configuration");
    return MPG_OK;
}

extern "C" int mpg_step_begin(mpg_train_ctx_t* c, int iteration, mpg_stream_t s) {
    MPG_REQUIRE(ctx_ok(c), "mpg_step_begin: incomplete context");
    const Layout l = layout(c);
    const int od = c->cfg.obs_dim, ad = c->cfg.act_dim, kind = c->cfg.env_kind;
    const float* policy = c->params + l.off[l.n_nets - 1];
    const float* policy_t = c->targets + l.off[l.n_nets - 1];
    if (!is_mpg(c)) {        // ---- NADP / TD3: sample, add, replay, gradients (optimizer.py:330-353) ----
        if (iteration % c->sampling_interval == 0) TRY(sample_and_add(c, policy, s));
        MPG_REQUIRE(c->ring_size > 0, "mpg_step_begin: empty replay ring");
        c->replay_times++;
        if (c->learner_version == 4 && c->prioritized) {
            TRY(mpg_per_sample_gather(c->per_sum, c->per_min, c->per_capacity, c->ring_size, c->batch, nullptr, c->replay_seed,
                                      c->replay_times, c->per_beta, c->idx, c->b_weights, od, ad, c->ring_obs, c->ring_act, c->ring_rew,
                                      c->ring_obs2, c->ring_done, c->b_obs, c->b_act, c->b_rew, c->b_obs2, c->b_done, s));
        } else {
            TRY(mpg_replay_sample_uniform(c->ring_size, c->batch, c->replay_seed, c->replay_times, od, ad, c->ring_obs, c->ring_act,
                                          c->ring_rew, c->ring_obs2, c->ring_done, c->idx, c->b_obs, c->b_act, c->b_rew, c->b_obs2,
                                          c->b_done, s));
        }
        c->learner_counter++;
        return c->learner_version == 3 ? nadp_gradients(c, l, s) : td3_gradients(c, l, s);
    }
    // ---- worker.sample + replay_buffer.add_batch (optimizer.py:332-337, worker.py:91-119) ----
    // MPG-v2 draws its minibatch right after the add: the last env launch gathers it in spare workgroups (the random ring
    // reads overlap the env's sub-steps) except the rows drawn from the slots that launch is writing
    const bool draws_now = c->learner_version == 2 && c->learner_counter % c->num_batch_reuse == 0;
    bool pre_gathered = false;
    int fresh_start = 0;
    if (iteration % c->sampling_interval == 0) {
        for (int it = 0; it < c->sample_iters; ++it) {
            MPG_REQUIRE(c->num_agent <= c->ring_capacity, "mpg_step_begin: ring smaller than one sample");
            const bool predraw = draws_now && it == c->sample_iters - 1 && kind == MPG_ENV_PATH_TRACKING && od == 6;
            mpg_replay_draw_t pd = {};
            if (predraw) {
                pd.n_storage = std::min(c->ring_size + c->num_agent, c->ring_capacity);
                pd.seed = c->replay_seed; pd.ctr = c->replay_times + 1;
                pd.idx_out = c->idx; pd.done_out = c->b_done;
                pre_gathered = true;
                fresh_start = c->ring_next;
            }
            if (worker_step_fused(c)) {
                // the policy pass, env.step -> ring slot (next + i) % capacity -> env.reset of the done agents (and the draw): one launch
                TRY(mpg_worker_step(&c->cfg, policy, c->num_agent, c->env_state, c->w_obs, c->explore_sigma, c->worker_seed, c->noise_ctr++,
                                    c->w_act, c->ring_capacity, c->ring_next, c->ring_obs, c->ring_act, c->ring_rew, c->ring_obs2,
                                    c->ring_done, c->env_seed, c->env_ctr++, c->w_done, predraw ? &pd : nullptr, c->batch, c->b_obs, c->b_act,
                                    c->b_rew, c->b_obs2, s));
            } else {
                TRY(mpg_policy_action(&c->cfg, policy, c->num_agent, c->w_obs, c->explore_sigma, c->worker_seed, c->noise_ctr++,
                                      c->w_act, s));
                // env.step -> ring slot (next + i) % capacity -> env.reset of the done agents, one launch
                mpg_prof_begin(c->cfg.prof, 2, mpg_stream(s));
                if (predraw) {
                    TRY(mpg_env_step_store_reset_draw(kind, c->num_agent, od, c->env_state, c->w_act, c->ring_capacity, c->ring_next,
                                                      c->ring_obs, c->ring_act, c->ring_rew, c->ring_obs2, c->ring_done, c->env_seed,
                                                      c->env_ctr++, c->w_obs, c->w_done, &pd, c->batch, c->b_obs, c->b_act, c->b_rew,
                                                      c->b_obs2, s));
                } else {
                    TRY(mpg_env_step_store_reset(kind, c->num_agent, od, c->env_state, c->w_act, c->ring_capacity, c->ring_next,
                                                 c->ring_obs, c->ring_act, c->ring_rew, c->ring_obs2, c->ring_done, c->env_seed,
                                                 c->env_ctr++, c->w_obs, c->w_done, s));
                }
                mpg_prof_end(c->cfg.prof, 2, mpg_stream(s));
            }
            c->ring_next = (c->ring_next + c->num_agent) % c->ring_capacity;
            c->ring_size = std::min(c->ring_size + c->num_agent, c->ring_capacity);
        }
    }
    // ---- replay_buffer.replay (optimizer.py:340-341; buffer.py:70-91) ----
    MPG_REQUIRE(c->ring_size > 0, "mpg_step_begin: empty replay ring");
    c->replay_times++;
    mpg_replay_draw_t draw = {};
    bool draw_in_gradients = false;
    if (c->learner_counter % c->num_batch_reuse == 0) {       // get_batch_data, mpg_learner.py:402-403
        if (c->learner_version == 2) {
            // the minibatch draw and the clipped double-Q target both ride in the first launch of mpg_mpg_gradients
            draw.n_storage = c->ring_size; draw.seed = c->replay_seed; draw.ctr = c->replay_times;
            draw.ring_obs = c->ring_obs; draw.ring_act = c->ring_act; draw.ring_rew = c->ring_rew; draw.ring_obs2 = c->ring_obs2;
            draw.ring_done = c->ring_done; draw.idx_out = c->idx; draw.done_out = c->b_done;
            draw.pre_gathered = pre_gathered ? 1 : 0; draw.capacity = c->ring_capacity; draw.fresh_start = fresh_start;
            draw.fresh_count = pre_gathered ? c->num_agent : 0;
            draw_in_gradients = true;
        } else {   // MPGLearner.sample + compute_n_step_target, mpg_learner.py:109-124,146-169
            TRY(mpg_replay_sample_uniform(c->ring_size, c->batch, c->replay_seed, c->replay_times, od, ad, c->ring_obs, c->ring_act,
                                          c->ring_rew, c->ring_obs2, c->ring_done, c->idx, c->b_obs, c->b_act, c->b_rew, c->b_obs2,
                                          c->b_done, s));
            MPG_REQUIRE(c->l_env_state && c->l_obs && c->l_act && c->l_rewards && c->l_done, "mpg_step_begin: MPG-v1 needs the learner env buffers");
            TRY(mpg_env_reset_from_obs(kind, c->batch, od, c->l_env_state, c->b_obs, s));
            for (int t = 0; t < c->n; ++t) {
                const float* act = c->b_act;
                if (t > 0) {
                    TRY(mpg_policy_action(&c->cfg, policy, c->batch, c->l_obs, 0.f, 0, 0, c->l_act, s));
                    act = c->l_act;
                }
                TRY(mpg_env_step(kind, c->batch, od, c->l_env_state, act, c->l_obs, c->l_rewards + (size_t)t * c->batch, c->l_done,
                                 c->l_done_intended, s));
            }
            TRY(mpg_nstep_targets(&c->cfg, policy_t, c->targets + l.off[0], c->batch, c->n, c->l_rewards, c->l_obs, c->b_targets,
                                  c->ws0, c->ws0_bytes, s));
        }
    }
    c->learner_counter++;
    // ---- learner.compute_gradient (mpg_learner.py:401-431), un-clipped partials scaled by 1/B_global ----
    const float inv_b = 1.f / ((float)c->batch * (float)c->world_size);
    float w[4];
    rule_based_weights(iteration, c->total_ite, c->eta, c->select, c->n_select, w);
    // MPG-v2 with num_batch_reuse > 1 keeps the targets of the batch they were computed for (mpg_learner.py:402-403)
    const bool fresh = (c->learner_counter - 1) % c->num_batch_reuse == 0;
    const float* y_in = (c->learner_version == 2 && fresh) ? nullptr : c->b_targets;
    // scheduling option (include/mpg_hip.h, mpg_grad_opts_t): with an exchange and the caller's event, the critics' gradient is finished
    // ahead of the reverse sweep
    c->grad_opts.critics_ready_event = exchanged(c) ? c->critics_ready_event : nullptr;
    c->cfg.grad_opts = &c->grad_opts;
    const int rc = mpg_mpg_gradients(&c->cfg, l.n_nets - 1, c->params, c->targets, c->batch, c->b_obs, c->b_act, c->b_rew, c->b_obs2, y_in,
                                     c->M, c->n, c->select, c->n_select, w, nullptr, c->learner_seed, c->learner_counter, inv_b, c->grad,
                                     c->grad + l.n_grad, c->b_targets, exchanged(c) ? nullptr : c->clip_scratch,
                                     draw_in_gradients ? &draw : nullptr, c->ws1, c->ws1_bytes, s);
    c->cfg.grad_opts = nullptr;
    return rc;
}

extern "C" int mpg_step_end(mpg_train_ctx_t* c, int iteration, mpg_stream_t s) {
    MPG_REQUIRE(ctx_ok(c) && c->norms && c->nonfinite && c->adam_m && c->adam_v && c->clip_scratch, "mpg_step_end: incomplete context");
    const Layout l = layout(c);
    // per-network clip_by_global_norm (mpg_learner.py:415-431): on one GPU the partial sums of squares were left in
    // clip_scratch by mpg_mpg_gradients; after an all-reduce they are recomputed from the reduced gradient
    // (NADP / TD3: their gradient launches leave no partials, they are always taken from the gradient buffer); an exchange that
    // sums with mpg_sum_slots_sq has left them already (clip_partials_ready)
    if ((exchanged(c) && !c->clip_partials_ready) || (!exchanged(c) && !is_mpg(c))) TRY(mpg_sq_partials(c->grad, l.sizes, l.n_nets, c->clip_scratch, s));
    // PolicyWithQs.apply_gradients, policy.py:123-156
    const bool delayed = iteration % c->delay_update == 0;
    float lr_t[3];
    int do_adam[3], do_polyak[3];
    for (int k = 0; k < l.n_nets; ++k) {
        const bool is_policy = k == l.n_nets - 1;
        const bool upd = !is_policy || delayed;
        const long long t = c->opt_steps[k] + 1;
        lr_t[k] = adam_step_size(is_policy ? c->policy_lr : c->value_lr, c->opt_steps[k]);
        do_adam[k] = upd ? 1 : 0;
        do_polyak[k] = delayed ? 1 : 0;
        if (upd) c->opt_steps[k] = t;
    }
    mpg_prof_begin(c->cfg.prof, 9, mpg_stream(s));
    const int rc = mpg_clip_adam_polyak(c->params, c->adam_m, c->adam_v, c->targets, c->grad, c->clip_scratch, l.sizes, l.n_nets, c->clip,
                                        lr_t, do_adam, do_polyak, c->tau, c->norms, c->nonfinite, c->cfg.wcache[0], c->cfg.wcache[1], s);
    mpg_prof_end(c->cfg.prof, 9, mpg_stream(s));
    return rc;
}
