// K2+K3+K4: fused n-step model rollout + return accumulation (forward sweep) and its reverse sweep (mixed policy
// gradient) for gfx950.
//
// A workgroup owns 16 trajectories for the WHOLE horizon: the policy's 256x256 kernel stays in registers across all
// n+1 policy evaluations (mlp_core.h), the vehicle / pendulum state of a trajectory stays in the registers of one
// lane, and only the hidden activations needed by the reverse sweep are streamed to HBM (G16 layout, 1 KiB coalesced
// per wave instruction).  No inter-workgroup communication.
//
// Reference: MPGLearner.model_rollout_for_policy_update / policy_forward_and_backward
// (learners/mpg_learner.py:226-286, :356-365), PathTrackingModel.rollout_out (envs_and_models/path_tracking_env.py:
// 279-297, f_xu :78-138, rewards :181-199), InvertedPendulumModel (envs_and_models/inverted_pendulum_model.py:16-97),
// NADPLearner (learners/nadp.py:87-194).  The reverse sweep replaces tf.GradientTape; its closed-form model adjoints
// are pinned against autograd in tests/test_model_vjp.py.
//
// This translation unit holds the entry points and the small kernels; the two sweeps are rollout_fwd.hip / rollout_bwd.hip.
#include "rollout_common.h"

using namespace mlp;
using namespace rollout;

namespace {

// ---------------------------------------------------------------------------------------------------------------
// returns, statistics and the critic-side seeds of the reverse sweep
// ---------------------------------------------------------------------------------------------------------------
// per trajectory: ret_k = G_k + gamma^k * Q_k.  M-mean over the tiles, then over this GPU's rows: sum and sum of
// squares per slice (mpg_learner.py:266-274).  dyq[k][r] = dL/dQ_k = -w_k * gamma^k * inv_b_global / M.
struct RetCoef {
    float gpow[MAXSEL];   // gamma^k
    float coef[MAXSEL];   // -w_k * gamma^k * inv_b_global / M
};
__global__ void __launch_bounds__(1024) k_returns(int rows, int M, int n_sel, const RetCoef rc,
                                                  const float* __restrict__ Q, const float* __restrict__ GK,
                                                  float* __restrict__ dyq, float* __restrict__ ret_sum,
                                                  float* __restrict__ ret_sqsum) {
    __shared__ float red[2][1024];
    const long R = (long)rows * M;
    for (int k = 0; k < n_sel; ++k) {
        float s = 0.f, s2 = 0.f;
        if (M == 1) {
            // eight row blocks of a thread at a time: their sixteen loads are requested together (unconditional, from clamped rows) and
            // then summed in the order of the plain loop - one memory round trip where the plain loop makes eight, one after the other
            for (int b0 = threadIdx.x; b0 < rows; b0 += 8 * 1024) {
                float gk[8], qv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int b = b0 + j * 1024 < rows ? b0 + j * 1024 : b0;
                    gk[j] = GK[k * R + b];
                    qv[j] = Q[k * R + b];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int b = b0 + j * 1024;
                    if (b < rows) {
                        float m = 0.f;
                        m += gk[j] + rc.gpow[k] * qv[j];
                        m /= 1.f;
                        s += m;
                        s2 += m * m;
                        dyq[k * R + b] = rc.coef[k];
                    }
                }
            }
        } else
        for (int b = threadIdx.x; b < rows; b += 1024) {
            float m = 0.f;
            for (int mm = 0; mm < M; ++mm) {
                const long tr = (long)mm * rows + b;
                m += GK[k * R + tr] + rc.gpow[k] * Q[k * R + tr];
                dyq[k * R + tr] = rc.coef[k];
            }
            m /= (float)M;
            s += m;
            s2 += m * m;
        }
        red[0][threadIdx.x] = s;
        red[1][threadIdx.x] = s2;
        __syncthreads();
        for (int w = 512; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) {
                red[0][threadIdx.x] += red[0][threadIdx.x + w];
                red[1][threadIdx.x] += red[1][threadIdx.x + w];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            ret_sum[k] = red[0][0];
            ret_sqsum[k] = red[1][0];
        }
        __syncthreads();
    }
}

// y = G_n + gamma^n * Q   (nadp.py:117-126)
__global__ void k_gq(int n, const float* __restrict__ G, const float* __restrict__ Q, float gpow, float* __restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = G[i] + gpow * Q[i];
}

// y[k][b] = mean over the M copies of G_k + gamma^k * Q_k, Q clipped to [-0.5, 0] where `clip[k]`   (mpg_learner.py:202-216)
struct QestCoef {
    float gpow[MAXSEL];
    int clip[MAXSEL];
};
__global__ void k_qest(int rows, int M, int n_sel, const QestCoef qc, const float* __restrict__ Q,
                       const float* __restrict__ GK, float* __restrict__ y) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= rows) return;
    const long R = (long)rows * M;
    for (int k = 0; k < n_sel; ++k) {
        float m = 0.f;
        for (int mm = 0; mm < M; ++mm) {
            const long tr = (long)mm * rows + b;
            float q = Q[k * R + tr];
            if (qc.clip[k]) q = fminf(fmaxf(q, -0.5f), 0.f);
            m += GK[k * R + tr] + qc.gpow[k] * q;
        }
        y[(long)k * rows + b] = m / (float)M;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
inline char* align256(char* p) { return reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 255) & ~uintptr_t(255)); }
struct Carver {
    char *p, *end;
    Carver(void* ws, size_t bytes) : p(align256((char*)ws)), end((char*)ws + bytes) {}
    float* take(size_t nfloat) {
        float* r = reinterpret_cast<float*>(p);
        p = align256(p + nfloat * sizeof(float));
        return r;
    }
};
inline size_t pad256(size_t nfloat) { return ((nfloat * sizeof(float) + 255) & ~size_t(255)) + 256; }

inline bool cfg_ok(const mpg_cfg_t* c) {
    return c && ((c->obs_dim >= 6 && c->obs_dim <= 16 && c->act_dim == 2 && c->env_kind == MPG_ENV_PATH_TRACKING) ||
                 (c->obs_dim == 4 && c->act_dim == 1 && c->env_kind == MPG_ENV_INVERTED_PENDULUM)) &&
           !(c->policy_out_act == MPG_ACT_TANH && c->action_range > 0.f);   // see learner_api.hip:cfg_ok
}


struct PgLayout {
    size_t h, sa, xq, gk, q, dyq, hq, gxq, dz, dz3, slabs, small, thin, xw, total;
};

// the thin gradients ride in the reverse sweep (rollout_common.h) when every step is differentiated through the parameters, the
// trajectories are the batch rows and the packed backward image exists (the THIN instantiations are packed-image kernels)
inline bool thin_in_sweep(const mpg_cfg_t* cfg, int M, int stash_all) { return stash_all && M == 1 && cfg->obs_dim <= 6; }
// observations with look-ahead entries: the first layer's inputs of the stashed steps are written out for the weight-gradient launch
// (k_wide_inputs) unless they are the caller's batch itself (M == 1, step 0 only)
inline bool wide_inputs_needed(const mpg_cfg_t* cfg, int M, int stash_all) { return cfg->obs_dim > 6 && (M > 1 || stash_all); }

// out [T][R][od] (raw, the scale is applied by the consumer): step 0 - the caller's batch row of the trajectory (look-ahead entries
// computed by the env); steps t > 0 - the model state's base entries and, for every look-ahead entry, a copy of entry fut_src
// (PathTrackingModel._get_obs, path_tracking_env.py:262-268)
__global__ void k_wide_inputs(int T, long R, int rows, int od, int obs_base, int fut_src, const float* __restrict__ obs0,
                              const float* __restrict__ SA, float* __restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)T * R * od) return;
    const int i = (int)(idx % od);
    const long tr = (idx / od) % R, t = idx / od / R;
    float v;
    if (t == 0) v = obs0[(tr % rows) * od + i];
    else v = SA[(t * R + tr) * SAW + (i < obs_base ? i : fut_src)];
    out[idx] = v;
}

PgLayout pg_layout(const mpg_cfg_t* cfg, int rows, int M, int n, int n_sel, int stash_all) {
    PgLayout l;
    const long R = (long)rows * M;
    const int qin = cfg->obs_dim + cfg->act_dim;
    const int T = stash_all ? n + 1 : 1;
    l.h = (size_t)(n + 1) * stash_floats(R);
    l.sa = (size_t)(n + 1) * R * SAW;
    l.xq = (size_t)n_sel * R * qin;
    l.gk = l.q = l.dyq = (size_t)n_sel * R;
    l.hq = stash_floats(n_sel * R);
    l.gxq = (size_t)n_sel * R * qin;
    l.dz = (size_t)T * stash_floats(R);
    l.dz3 = (size_t)T * R * cfg->act_dim;
    l.slabs = wgrad_workspace_floats((int)(T * stash_floats(R) / H), cfg->obs_dim, 2 * cfg->act_dim);
    l.small = 64;
    l.thin = thin_in_sweep(cfg, M, stash_all) ? (size_t)256 * thin_floats(cfg->obs_dim, 2 * cfg->act_dim) : 0;
    l.xw = wide_inputs_needed(cfg, M, stash_all) ? (size_t)T * R * cfg->obs_dim : 0;
    l.total = 2 * pad256(l.h) + pad256(l.sa) + pad256(l.xq) + 3 * pad256(l.gk) + 2 * pad256(l.hq) + pad256(l.gxq) +
              2 * pad256(l.dz) + pad256(l.dz3) + pad256(l.slabs) + 2 * pad256(l.small) + pad256(l.thin) + pad256(l.xw);
    return l;
}


// ---- launch helpers shared by mpg_rollout_pg and mpg_mpg_gradients ---------------------------------------------------
struct Coefs {
    float gpow[MAXSEL], coef[MAXSEL], rho[MAXN];
};

// gamma^k, dL/dQ_k and dL/d(raw reward_t) for the loss sum_k w_k * (-mean return_k)   (mpg_learner.py:251,360-361)
Coefs make_coefs(const mpg_cfg_t* cfg, const int* select, int n_select, const float* w, float inv_b_global, int M) {
    Coefs c;
    for (int k = 0; k < MAXSEL; ++k) c.gpow[k] = c.coef[k] = 0.f;
    for (int t = 0; t < MAXN; ++t) c.rho[t] = 0.f;
    const float cc = inv_b_global / (float)M;
    for (int k = 0; k < n_select; ++k) {
        c.gpow[k] = powf(cfg->gamma, (float)select[k]);                   // tf.pow(gamma, k) in float32
        c.coef[k] = -w[k] * c.gpow[k] * cc;
        for (int t = 0; t < select[k]; ++t) c.rho[t] += -w[k] * cc * powf(cfg->gamma, (float)t) * cfg->rew_scale;
    }
    return c;
}

int run_rollout_fwd(const mpg_cfg_t* cfg, const float* policy_params, int rows, int M, int n, const int* select, int n_select,
                    const float* obs0, const float* eps, uint64_t noise_seed, uint64_t noise_ctr, float* H1, float* H2,
                    float* SA, float* XQ, float* GK, hipStream_t s) {
    const long R = (long)rows * M;
    // ---- forward sweep ----
    RollArgs fa;
    fill_roll(fa, cfg, policy_params, rows, M, n);
    fa.obs0 = obs0; fa.act0 = nullptr; fa.eps = eps; fa.H1 = H1; fa.H2 = H2; fa.SA = SA;
    fa.nk0 = (uint32_t)noise_seed; fa.nk1 = (uint32_t)(noise_seed >> 32); fa.nc0 = (uint32_t)noise_ctr; fa.nc1 = (uint32_t)(noise_ctr >> 32);
    fa.n_sel = n_select;
    for (int k = 0; k < MAXSEL; ++k) fa.sel[k] = k < n_select ? select[k] : -1;
    fa.XQ = XQ; fa.GK = GK;
    const long ngroups = (R + GROUP - 1) / GROUP;
    return launch_rollout_fwd(fa, cfg->env_kind, ngroups, n, s, cfg->prof);
}

int run_rollout_bwd(const mpg_cfg_t* cfg, const float* policy_params, int rows, int M, int n, const int* select, int n_select,
                    const float* rho, const float* H1, const float* H2, const float* SA, const float* GXQ,
                    int all_steps_param_grad, float* DZ1, float* DZ2, float* DZ3, hipStream_t s, float* thin_part = nullptr) {
    const long R = (long)rows * M;
    const long ngroups = (R + GROUP - 1) / GROUP;
    const int od = cfg->obs_dim, ad = cfg->act_dim;
    RollArgs fa;
    fill_roll(fa, cfg, policy_params, rows, M, n);
    for (int k = 0; k < MAXSEL; ++k) fa.sel[k] = k < n_select ? select[k] : -1;
    // ---- reverse sweep ----
    RollBwdArgs ba;
    ba.policy = policy_params; ba.rows = rows; ba.M = M; ba.n = n;
    ba.obs_dim = fa.obs_dim;
    for (int i = 0; i < 16; ++i) ba.obs_scale[i] = fa.obs_scale[i];
    ba.out_tanh = fa.out_tanh; ba.out_scale = fa.out_scale; ba.inv_out_scale = 1.f / fa.out_scale;
    ba.H1 = H1; ba.H2 = H2; ba.SA = SA; ba.n_sel = n_select;
    for (int k = 0; k < MAXSEL; ++k) ba.sel[k] = fa.sel[k];
    ba.GXQ = GXQ;
    for (int t = 0; t < MAXN; ++t) ba.rho[t] = rho[t];
    ba.stash_all = all_steps_param_grad ? 1 : 0;
    ba.DZ1 = DZ1; ba.DZ2 = DZ2; ba.DZ3 = DZ3;
    ba.pack = weight_cache_lookup(cfg, make_net(policy_params, od, 2 * ad).W2, 1);
    ba.thin_part = thin_part;
    return launch_rollout_bwd(ba, cfg->env_kind, ngroups, n, s, cfg->prof);
}

}  // namespace

extern "C" size_t mpg_rollout_pg_workspace_bytes(const mpg_cfg_t* cfg, int rows, int M, int n, int n_select,
                                                 int all_steps_param_grad) {
    if (!cfg_ok(cfg) || rows <= 0 || M <= 0 || n <= 0 || n >= MAXN || n_select <= 0 || n_select > MAXSEL) return 0;
    return pg_layout(cfg, rows, M, n, n_select, all_steps_param_grad).total;
}

extern "C" int mpg_rollout_pg(const mpg_cfg_t* cfg, const float* policy_params, const float* q1_params, int rows, int M,
                              int n, const int* select, int n_select, const float* w, const float* obs0, const float* eps,
                              uint64_t noise_seed, uint64_t noise_ctr, float inv_b_global, int all_steps_param_grad, float* ret_sum, float* ret_sqsum, float* grad,
                              void* ws, size_t ws_bytes, mpg_stream_t stream) {
    MPG_REQUIRE(cfg_ok(cfg), "mpg_rollout_pg: unsupported cfg (obs/act dims, env_kind)");
    MPG_REQUIRE(policy_params && q1_params && select && w && obs0 && ret_sum && ret_sqsum && grad && ws,
                "mpg_rollout_pg: null pointer");
    MPG_REQUIRE(rows > 0 && M > 0 && n > 0 && n < MAXN && n_select > 0 && n_select <= MAXSEL, "mpg_rollout_pg: bad sizes");
    const long R = (long)rows * M;
    MPG_REQUIRE(!all_steps_param_grad || R % GROUP == 0, "mpg_rollout_pg: all_steps_param_grad needs rows*M %% 16 == 0");
    for (int k = 0; k < n_select; ++k) MPG_REQUIRE(select[k] >= 0 && select[k] <= n, "mpg_rollout_pg: slice out of range");
    const PgLayout l = pg_layout(cfg, rows, M, n, n_select, all_steps_param_grad);
    if (ws_bytes < l.total) {
        mpg_set_error("mpg_rollout_pg: workspace too small (%zu < %zu)", ws_bytes, l.total);
        return MPG_EWORKSPACE;
    }
    hipStream_t s = mpg_stream(stream);
    const int od = cfg->obs_dim, ad = cfg->act_dim, qin = od + ad;
    Carver cv(ws, ws_bytes);
    float* H1 = cv.take(l.h); float* H2 = cv.take(l.h);
    float* SA = cv.take(l.sa); float* XQ = cv.take(l.xq);
    float* GK = cv.take(l.gk); float* Q = cv.take(l.q); float* DYQ = cv.take(l.dyq);
    float* HQ1 = cv.take(l.hq); float* HQ2 = cv.take(l.hq); float* GXQ = cv.take(l.gxq);
    float* DZ1 = cv.take(l.dz); float* DZ2 = cv.take(l.dz); float* DZ3 = cv.take(l.dz3);
    float* slabs = cv.take(l.slabs);
    float* thin_part = l.thin ? cv.take(l.thin) : nullptr;
    float* XW = l.xw ? cv.take(l.xw) : nullptr;
    // (the THIN reverse sweep is a packed-image kernel: without the caller's weight cache the thin parts stay in the wgrad launch)
    if (thin_part && !weight_cache_lookup(cfg, make_net(policy_params, cfg->obs_dim, 2 * cfg->act_dim).W2, 1)) thin_part = nullptr;

    // ---- forward sweep ----
    int rc = run_rollout_fwd(cfg, policy_params, rows, M, n, select, n_select, obs0, eps, noise_seed, noise_ctr, H1, H2, SA, XQ,
                             GK, s);
    if (rc) return rc;
    const long ngroups = (R + GROUP - 1) / GROUP;
    (void)ngroups;

    // ---- critic at the selected slices: values, returns, input gradients ----
    OutSpec lin; lin.out_tanh = 0; lin.out_scale = 1.f; lin.sigma = 0.f; lin.seed = lin.ctr = 0;
    const int RQ = (int)(n_select * R);
    rc = launch_forward(cfg, q1_params, qin, 1, 1, RQ, xspec(XQ, qin, nullptr, 0, nullptr, 0), lin, Q, 1, HQ1, HQ2, s);
    if (rc) return rc;
    const Coefs cf = make_coefs(cfg, select, n_select, w, inv_b_global, M);
    RetCoef rcf;
    for (int k = 0; k < MAXSEL; ++k) { rcf.gpow[k] = cf.gpow[k]; rcf.coef[k] = cf.coef[k]; }
    hipLaunchKernelGGL(k_returns, dim3(1), dim3(1024), 0, s, rows, M, n_select, rcf, Q, GK, DYQ, ret_sum, ret_sqsum);
    MPG_CHECK_LAUNCH("k_returns");
    rc = launch_backward(cfg, q1_params, qin, 1, 1, RQ, DYQ, 1, nullptr, 0, 0, 1.f, HQ1, HQ2, nullptr, nullptr, nullptr, GXQ, qin, s);
    if (rc) return rc;

    // ---- reverse sweep ----
    rc = run_rollout_bwd(cfg, policy_params, rows, M, n, select, n_select, cf.rho, H1, H2, SA, GXQ, all_steps_param_grad, DZ1, DZ2,
                         DZ3, s, thin_part);
    if (rc) return rc;

    // ---- policy weight gradient from the stashes (step 0 only, or every step for NADP) ----
    const int T = all_steps_param_grad ? n + 1 : 1;
    // the first layer's input of every stashed step: the (obs | action) records hold the six base entries; with look-ahead entries
    // (obs_dim > 6) it is the caller's batch itself (M == 1, step 0 only) or written out by k_wide_inputs
    XSpec xs = xspec(SA, od, nullptr, 0, cfg->obs_scale, od);
    xs.ld0 = SAW;
    if (od > 6) xs = xspec(obs0, od, nullptr, 0, cfg->obs_scale, od);
    if (XW) {
        const long nx = (long)T * R * od;
        hipLaunchKernelGGL(k_wide_inputs, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, s, T, R, rows, od, PathTracking::OBS,
                           PathTracking::FUT_SRC, obs0, SA, XW);
        MPG_CHECK_LAUNCH("k_wide_inputs");
        xs = xspec(XW, od, nullptr, 0, cfg->obs_scale, od);
    }
    // (with thin_part: dW2 from the chunk slabs, the thin parts from the sweep's per-workgroup partials, summed in one launch)
    const int n_part = (int)std::min<long>(256, (R + GROUP - 1) / GROUP);
    return launch_wgrad(cfg, od, 2 * ad, ad, (int)(T * R), xs, H1, H2, DZ1, DZ2, DZ3, inv_b_global / (float)M, grad, slabs, s,
                        thin_part != nullptr, thin_part, thin_part ? n_part : 0);
}

extern "C" size_t mpg_rollout_q_target_workspace_bytes(const mpg_cfg_t* cfg, int rows) {
    if (!cfg_ok(cfg) || rows <= 0) return 0;
    return pad256((size_t)rows * (cfg->obs_dim + cfg->act_dim)) + 2 * pad256(rows);
}

extern "C" int mpg_rollout_q_target(const mpg_cfg_t* cfg, const float* policy_params, const float* q1t, int rows, int n,
                                    const float* obs0, const float* act0, const float* eps, uint64_t noise_seed,
                                    uint64_t noise_ctr, float* y, void* ws, size_t ws_bytes, mpg_stream_t stream) {
    MPG_REQUIRE(cfg_ok(cfg) && policy_params && q1t && obs0 && act0 && y && ws && rows > 0 && n > 0 && n < MAXN,
                "mpg_rollout_q_target: bad argument");
    if (ws_bytes < mpg_rollout_q_target_workspace_bytes(cfg, rows)) {
        mpg_set_error("mpg_rollout_q_target: workspace too small");
        return MPG_EWORKSPACE;
    }
    hipStream_t s = mpg_stream(stream);
    const int qin = cfg->obs_dim + cfg->act_dim;
    Carver cv(ws, ws_bytes);
    float* XQ = cv.take((size_t)rows * qin); float* GK = cv.take(rows); float* Q = cv.take(rows);
    RollArgs fa;
    fill_roll(fa, cfg, policy_params, rows, 1, n);
    fa.obs0 = obs0; fa.act0 = act0; fa.eps = eps; fa.H1 = fa.H2 = nullptr; fa.SA = nullptr; fa.dbg = nullptr;
    fa.nk0 = (uint32_t)noise_seed; fa.nk1 = (uint32_t)(noise_seed >> 32); fa.nc0 = (uint32_t)noise_ctr; fa.nc1 = (uint32_t)(noise_ctr >> 32);
    fa.n_sel = 1;
    for (int k = 0; k < MAXSEL; ++k) fa.sel[k] = k == 0 ? n : -1;
    fa.XQ = XQ; fa.GK = GK;
    const long ngroups = (rows + GROUP - 1) / GROUP;
    {
        int rcq = launch_rollout_fwd(fa, cfg->env_kind, ngroups, n, s, nullptr);
        if (rcq) return rcq;
    }
    OutSpec lin; lin.out_tanh = 0; lin.out_scale = 1.f; lin.sigma = 0.f; lin.seed = lin.ctr = 0;
    int rc = launch_forward(cfg, q1t, qin, 1, 1, rows, xspec(XQ, qin, nullptr, 0, nullptr, 0), lin, Q, 1, nullptr, nullptr, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_gq, dim3((rows + 255) / 256), dim3(256), 0, s, rows, GK, Q, powf(cfg->gamma, (float)n), y);
    MPG_CHECK_LAUNCH("k_gq");
    return MPG_OK;
}

extern "C" size_t mpg_rollout_q_estimation_workspace_bytes(const mpg_cfg_t* cfg, int rows, int M, int n_select) {
    if (!cfg_ok(cfg) || rows <= 0 || M <= 0 || n_select <= 0 || n_select > MAXSEL) return 0;
    const size_t R = (size_t)rows * M;
    return pad256(n_select * R * (cfg->obs_dim + cfg->act_dim)) + 2 * pad256(n_select * R);
}

extern "C" int mpg_rollout_q_estimation(const mpg_cfg_t* cfg, const float* policy_params, const float* q1t, int rows, int M,
                                        const int* select, int n_select, const float* obs0, const float* act0, const float* eps,
                                        uint64_t noise_seed, uint64_t noise_ctr, float* y, void* ws, size_t ws_bytes,
                                        mpg_stream_t stream) {
    MPG_REQUIRE(cfg_ok(cfg) && policy_params && q1t && select && obs0 && act0 && y && ws && rows > 0 && M > 0 && n_select > 0 &&
                    n_select <= MAXSEL,
                "mpg_rollout_q_estimation: bad argument");
    int n = 0;
    for (int k = 0; k < n_select; ++k) {
        MPG_REQUIRE(select[k] >= 0 && select[k] < MAXN, "mpg_rollout_q_estimation: slice out of range");
        n = std::max(n, select[k]);
    }
    if (ws_bytes < mpg_rollout_q_estimation_workspace_bytes(cfg, rows, M, n_select)) {
        mpg_set_error("mpg_rollout_q_estimation: workspace too small");
        return MPG_EWORKSPACE;
    }
    hipStream_t s = mpg_stream(stream);
    const int qin = cfg->obs_dim + cfg->act_dim;
    const long R = (long)rows * M;
    Carver cv(ws, ws_bytes);
    float* XQ = cv.take((size_t)n_select * R * qin); float* GK = cv.take((size_t)n_select * R); float* Q = cv.take((size_t)n_select * R);
    RollArgs fa;
    fill_roll(fa, cfg, policy_params, rows, M, n);
    fa.obs0 = obs0; fa.act0 = act0; fa.eps = eps; fa.H1 = fa.H2 = nullptr; fa.SA = nullptr; fa.dbg = nullptr;
    fa.nk0 = (uint32_t)noise_seed; fa.nk1 = (uint32_t)(noise_seed >> 32); fa.nc0 = (uint32_t)noise_ctr; fa.nc1 = (uint32_t)(noise_ctr >> 32);
    fa.n_sel = n_select;
    for (int k = 0; k < MAXSEL; ++k) fa.sel[k] = k < n_select ? select[k] : -1;
    fa.XQ = XQ; fa.GK = GK;
    int rc = launch_rollout_fwd(fa, cfg->env_kind, (R + GROUP - 1) / GROUP, n, s, nullptr);
    if (rc) return rc;
    OutSpec lin; lin.out_tanh = 0; lin.out_scale = 1.f; lin.sigma = 0.f; lin.seed = lin.ctr = 0;
    rc = launch_forward(cfg, q1t, qin, 1, 1, (int)(n_select * R), xspec(XQ, qin, nullptr, 0, nullptr, 0), lin, Q, 1, nullptr, nullptr, s);
    if (rc) return rc;
    QestCoef qc;
    for (int k = 0; k < MAXSEL; ++k) {
        qc.gpow[k] = k < n_select ? powf(cfg->gamma, (float)select[k]) : 0.f;
        // the pendulum branch clips the bootstrap of every slice but the first (all_Qs[batch_size:], :206-209)
        qc.clip[k] = (k < n_select && cfg->env_kind == MPG_ENV_INVERTED_PENDULUM && select[k] >= 1) ? 1 : 0;
    }
    hipLaunchKernelGGL(k_qest, dim3((rows + 255) / 256), dim3(256), 0, s, rows, M, n_select, qc, Q, GK, y);
    MPG_CHECK_LAUNCH("k_qest");
    return MPG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// MPGLearner.compute_gradient minus the clip, as ONE entry point (mpg_learner.py:401-431): 7 launches
//   target (fused)  ->  critics fwd+err+bwd (fused)  ->  rollout forward  ->  critic at the slices fwd+bwd (fused)
//   ->  rollout reverse  ->  weight gradients of all networks  ->  slab + statistic reduction
// Falls back to the fine-grained entry points (same results up to summation order) when rows % 16 != 0 or M > 1.
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct MgLayout {
    size_t stash, dz3, loss_part, h, sa, xq, gk, gxq, ret_part, dz3p, slab_q, slab_p, fused_total, fallback0, fallback1;
};

MgLayout mg_layout(const mpg_cfg_t* cfg, int rows, int M, int n, int n_sel, int n_q) {
    MgLayout l;
    const long R = (long)rows * M;
    const int od = cfg->obs_dim, ad = cfg->act_dim, qin = od + ad;
    const long ngroups = (rows + GROUP - 1) / GROUP;
    l.stash = stash_floats(rows);
    l.dz3 = rows;
    l.loss_part = 2 * ngroups;
    l.h = (size_t)(n + 1) * stash_floats(R);
    l.sa = (size_t)(n + 1) * R * SAW;
    l.xq = l.gxq = (size_t)n_sel * R * qin;
    l.gk = (size_t)n_sel * R;
    l.ret_part = (size_t)n_sel * ((R + GROUP - 1) / GROUP) * 2;
    l.dz3p = (size_t)R * ad;
    l.slab_q = wgrad_workspace_floats(rows, qin, 1);
    l.slab_p = wgrad_workspace_floats((int)R, od, 2 * ad);
    l.fused_total = (size_t)n_q * (4 * pad256(l.stash) + pad256(l.dz3) + pad256(l.slab_q)) + pad256(l.loss_part) +
                    2 * pad256(l.h) + pad256(l.sa) + pad256(l.xq) + pad256(l.gk) + pad256(l.gxq) + pad256(l.ret_part) +
                    2 * pad256(stash_floats(R)) + pad256(l.dz3p) + pad256(l.slab_p) + pad256(2 * (size_t)rows);   // (+ the split target's Q values)
    l.fallback0 = std::max(mpg_q_targets_workspace_bytes(cfg, rows), mpg_q_loss_grad_workspace_bytes(cfg, rows));
    l.fallback1 = mpg_rollout_pg_workspace_bytes(cfg, rows, M, n, n_sel, 0);
    return l;
}

}  // namespace

extern "C" size_t mpg_mpg_gradients_workspace_bytes(const mpg_cfg_t* cfg, int rows, int M, int n, int n_select, int n_q) {
    if (!cfg_ok(cfg) || rows <= 0 || M <= 0 || n <= 0 || n >= MAXN || n_select <= 0 || n_select > MAXSEL || n_q < 1 || n_q > 2 ||
        n_q + 2 * n_select > 8)
        return 0;
    const MgLayout l = mg_layout(cfg, rows, M, n, n_select, n_q);
    return std::max(l.fused_total, l.fallback0 + l.fallback1 + 512);
}

extern "C" int mpg_mpg_gradients(const mpg_cfg_t* cfg, int n_q, const float* params, const float* target_params, int rows,
                                 float* obs, float* act, float* rew, float* obs_tp1,
                                 const float* y_in, int M, int n, const int* select, int n_select, const float* w,
                                 const float* eps, uint64_t noise_seed, uint64_t noise_ctr, float inv_b_global, float* grad,
                                 float* stats, float* y_out, float* sq_part, const mpg_replay_draw_t* draw, void* ws,
                                 size_t ws_bytes, mpg_stream_t stream) {
    MPG_REQUIRE(cfg_ok(cfg) && (n_q == 1 || n_q == 2), "mpg_mpg_gradients: unsupported cfg / n_q");
    MPG_REQUIRE(params && obs && act && select && w && grad && stats && y_out && ws, "mpg_mpg_gradients: null pointer");
    MPG_REQUIRE(y_in || (target_params && rew && obs_tp1), "mpg_mpg_gradients: either y_in or the target inputs are required");
    MPG_REQUIRE(rows > 0 && M > 0 && n > 0 && n < MAXN && n_select > 0 && n_select <= MAXSEL, "mpg_mpg_gradients: bad sizes");
    for (int k = 0; k < n_select; ++k) MPG_REQUIRE(select[k] >= 0 && select[k] <= n, "mpg_mpg_gradients: slice out of range");
    // the statistics block holds n_q losses + 2 sums per slice in 8 reduction jobs: checked before anything is enqueued
    MPG_REQUIRE(n_q + 2 * n_select <= 8, "mpg_mpg_gradients: too many statistics (n_select <= 3 with two critics)");
    // with critics_ready_event the gradient is produced in two parts for an exchange between GPUs: the clip partials belong behind
    // that exchange (mpg_sq_partials on the reduced buffer), so asking for them here is a caller error, not something to leave unwritten
    MPG_REQUIRE(!(cfg->grad_opts && cfg->grad_opts->critics_ready_event && sq_part),
                "mpg_mpg_gradients: sq_part must be null when grad_opts->critics_ready_event is set (take the clip partials after the exchange)");
    if (ws_bytes < mpg_mpg_gradients_workspace_bytes(cfg, rows, M, n, n_select, n_q)) {
        mpg_set_error("mpg_mpg_gradients: workspace too small");
        return MPG_EWORKSPACE;
    }
    hipStream_t s = mpg_stream(stream);
    const int od = cfg->obs_dim, ad = cfg->act_dim, qin = od + ad;
    const int q_size = net_size(qin, 1);
    const float* qp[2] = {params, n_q == 2 ? params + q_size : nullptr};
    const float* policy = params + (size_t)n_q * q_size;
    const float* qt[2] = {target_params, (target_params && n_q == 2) ? target_params + q_size : nullptr};
    const float* policy_t = target_params ? target_params + (size_t)n_q * q_size : nullptr;
    float* gq[2] = {grad, grad + q_size};
    float* gp = grad + (size_t)n_q * q_size;
    const MgLayout l = mg_layout(cfg, rows, M, n, n_select, n_q);
    const bool fused = rows % GROUP == 0 && M == 1 && (od == 6 || od == 4);      // the fused kernels are built for the base widths
    if (draw && (!fused || y_in)) {      // the draw cannot ride in the target launch: do it as its own launch
        MPG_REQUIRE(rew && obs_tp1, "mpg_mpg_gradients: a replay draw needs the rew / obs_tp1 output buffers");
        int rc = mpg_replay_sample_uniform(draw->n_storage, rows, draw->seed, draw->ctr, cfg->obs_dim, cfg->act_dim, draw->ring_obs,
                                           draw->ring_act, draw->ring_rew, draw->ring_obs2, draw->ring_done, draw->idx_out, obs, act,
                                           rew, obs_tp1, draw->done_out, stream);
        if (rc) return rc;
        draw = nullptr;
    }

    if (!fused) {   // ---- fallback: the fine-grained entry points ----
        char* w0 = align256((char*)ws);
        char* w1 = align256(w0 + l.fallback0);
        const float* y = y_in;
        if (!y) {
            int rc = mpg_q_targets(cfg, policy_t, qt[0], qt[1], rows, rew, obs_tp1, nullptr, 0.f, 0.f, y_out, w0, l.fallback0, stream);
            if (rc) return rc;
            y = y_out;
        }
        for (int k = 0; k < n_q; ++k) {
            int rc = mpg_q_loss_grad(cfg, qp[k], rows, obs, act, y, inv_b_global, stats + k, gq[k], nullptr, w0, l.fallback0, stream);
            if (rc) return rc;
        }
        // critics_ready_event (mpg_grad_opts_t): the critics' slice of `grad` and their losses are final here in this path too - a
        // caller that exchanges them on a second stream waits for THIS record, not for one left over from an earlier call
        if (cfg->grad_opts && cfg->grad_opts->critics_ready_event &&
            hipEventRecord(reinterpret_cast<hipEvent_t>(cfg->grad_opts->critics_ready_event), s) != hipSuccess) {
            mpg_set_error("mpg_mpg_gradients: hipEventRecord(critics_ready_event) failed");
            return MPG_EINVAL;
        }
        int rc = mpg_rollout_pg(cfg, policy, qp[0], rows, M, n, select, n_select, w, obs, eps, noise_seed, noise_ctr, inv_b_global, 0,
                                stats + 2, stats + 2 + n_select, gp, w1, l.fallback1, stream);
        if (rc || !sq_part) return rc;
        const int sizes[3] = {q_size, n_q == 2 ? q_size : net_size(od, 2 * ad), net_size(od, 2 * ad)};
        return mpg_sq_partials(grad, sizes, n_q + 1, sq_part, stream);
    }

    Carver cv(ws, ws_bytes);
    CriticStash st[2];
    float* slab_q[2] = {nullptr, nullptr};
    for (int k = 0; k < n_q; ++k) {
        st[k].h1 = cv.take(l.stash); st[k].h2 = cv.take(l.stash); st[k].dz1 = cv.take(l.stash); st[k].dz2 = cv.take(l.stash);
        st[k].dz3 = cv.take(l.dz3);
        slab_q[k] = cv.take(l.slab_q);
    }
    float* loss_part = cv.take(l.loss_part);
    float* H1 = cv.take(l.h); float* H2 = cv.take(l.h);
    float* SA = cv.take(l.sa); float* XQ = cv.take(l.xq); float* GK = cv.take(l.gk); float* GXQ = cv.take(l.gxq);
    float* ret_part = cv.take(l.ret_part);
    float* DZ1 = cv.take(stash_floats(rows)); float* DZ2 = cv.take(stash_floats(rows)); float* DZ3 = cv.take(l.dz3p);
    float* slab_p = cv.take(l.slab_p);
    float* qpart = cv.take(2 * (size_t)rows);

    // Split target: with both target critics and one pair of row groups per two CUs the target
    // launch would leave half of the chip idle (256 groups = 128 workgroups of two); instead workgroup (p, h) runs the target
    // policy and target critic h on pair p - 256 workgroups, two image loads and two passes each instead of three - and the
    // critic launch finishes y = r~ + gamma * min(Q1t, Q2t) (same arithmetic, bit-identical y) and writes y_out.
    const bool split = !y_in && n_q == 2 && n_select == 2 && qt[1] && rows / GROUP >= 256;
    const float* y = y_in;
    if (!y) {   // 1. clipped double-Q (or single-Q) target, mpg_learner.py:126-134
        const DrawOut dout{obs, act, rew, obs_tp1};
        int rc = launch_target_fused(cfg, policy_t, qt[0], qt[1], rows, rew, obs_tp1, nullptr, 0.f, 0.f, y_out, s, draw, &dout,
                                     split ? qpart : nullptr);
        if (rc) return rc;
        y = y_out;
    }
    int rc;
    const Coefs cf = make_coefs(cfg, select, n_select, w, inv_b_global, 1);
    if (n_select == 2) {
        // 2. rollout forward sweep
        rc = run_rollout_fwd(cfg, policy, rows, 1, n, select, n_select, obs, eps, noise_seed, noise_ctr, H1, H2, SA, XQ, GK, s);
        if (rc) return rc;
        // 3.+4. critics (forward, error, input-side backward, mpg_learner.py:326-354) and the critic at the two selected
        //       slices (returns and input gradients) in one launch
        rc = launch_critic_fused(cfg, qp, n_q, rows, obs, act, y, inv_b_global, st, loss_part, XQ, GK, cf.gpow, cf.coef, ret_part,
                                 GXQ, s, split ? qpart : nullptr, rew, y_out);
        if (rc) return rc;
    } else {
        // 2. critics: forward, error, input-side backward (mpg_learner.py:326-354)
        rc = launch_qloss_fused(cfg, qp, n_q, rows, obs, act, y, inv_b_global, st, loss_part, nullptr, s);
        if (rc) return rc;
        // 3. rollout forward sweep
        rc = run_rollout_fwd(cfg, policy, rows, 1, n, select, n_select, obs, eps, noise_seed, noise_ctr, H1, H2, SA, XQ, GK, s);
        if (rc) return rc;
        // 4. critic at the selected slices: returns and input gradients
        rc = launch_qslice_fused(cfg, qp[0], qin, rows, n_select, XQ, GK, cf.gpow, cf.coef, ret_part, GXQ, s);
        if (rc) return rc;
    }
    // 6./7. weight gradients of every network + all scalar statistics
    WgradJob jobs[3];
    const XSpec xq = xspec(obs, od, act, ad, cfg->obs_scale, od);
    for (int k = 0; k < n_q; ++k) {
        jobs[k].in_dim = qin; jobs[k].out_dim = 1; jobs[k].ou = 1; jobs[k].rows = rows; jobs[k].x = xq;
        jobs[k].h1 = st[k].h1; jobs[k].h2 = st[k].h2; jobs[k].dz1 = st[k].dz1; jobs[k].dz2 = st[k].dz2; jobs[k].dz3 = st[k].dz3;
        jobs[k].inv_b = inv_b_global;
        jobs[k].grad = gq[k]; jobs[k].slabs = slab_q[k];
    }
    WgradJob& jp = jobs[n_q];
    jp.in_dim = od; jp.out_dim = 2 * ad; jp.ou = ad; jp.rows = rows;
    jp.x = xspec(SA, od, nullptr, 0, cfg->obs_scale, od);
    jp.x.ld0 = SAW;
    jp.h1 = H1; jp.h2 = H2; jp.dz1 = DZ1; jp.dz2 = DZ2; jp.dz3 = DZ3; jp.inv_b = inv_b_global; jp.grad = gp; jp.slabs = slab_p;
    const int ngroups = rows / GROUP;
    SumJob sums[8];
    int ns = 0;
    for (int k = 0; k < n_q; ++k) { sums[ns].src = loss_part + (size_t)k * ngroups; sums[ns].n = ngroups; sums[ns].stride = 1; sums[ns].dst = stats + k; ++ns; }
    for (int k = 0; k < n_select && ns + 1 < 8; ++k) {
        sums[ns].src = ret_part + (size_t)k * ngroups * 2; sums[ns].n = ngroups; sums[ns].stride = 2; sums[ns].dst = stats + 2 + k; ++ns;
        sums[ns].src = ret_part + (size_t)k * ngroups * 2 + 1; sums[ns].n = ngroups; sums[ns].stride = 2; sums[ns].dst = stats + 2 + n_select + k; ++ns;
    }
    // Scheduling option (mpg_cfg_t.grad_opts, round 5):
    //   critics_ready_event - the critics' gradient is FINISHED (chunk products, slab sums, loss sums) right behind the critic launch and
    //                         the event recorded there, so that a caller can exchange it between GPUs under the reverse sweep; the
    //                         policy's follows the sweep as a second weight-gradient + reduction pair (one launch more; the critics'
    //                         chunk products ahead of the sweep cost ~9 us on one GPU, EXPERIMENTS.md "Why nothing overlaps the exchange");
    const mpg_grad_opts_t* opts = cfg->grad_opts;
    hipEvent_t critics_ready = opts ? reinterpret_cast<hipEvent_t>(opts->critics_ready_event) : nullptr;
    if (critics_ready) {
        // the loss sums are the first n_q scalar jobs; no clip partials: an exchanged gradient gets them afterwards (mpg_sq_partials)
        rc = launch_wgrad_multi(cfg, jobs, n_q, sums, n_q, nullptr, s);
        if (rc) return rc;
        if (hipEventRecord(critics_ready, s) != hipSuccess) { mpg_set_error("mpg_mpg_gradients: hipEventRecord(critics_ready_event) failed"); return MPG_EINVAL; }
    }
    // 5. reverse sweep
    rc = run_rollout_bwd(cfg, policy, rows, 1, n, select, n_select, cf.rho, H1, H2, SA, GXQ, 0, DZ1, DZ2, DZ3, s);
    if (rc) return rc;
    if (critics_ready) return launch_wgrad_multi(cfg, jobs + n_q, 1, sums + n_q, ns - n_q, nullptr, s);
    return launch_wgrad_multi(cfg, jobs, n_q + 1, sums, ns, sq_part, s);
}
