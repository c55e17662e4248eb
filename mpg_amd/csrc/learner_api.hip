// C-ABI entry points built from the generic network kernels: policy actions, critic targets, critic loss +
// gradient.  (The fused n-step rollout lives in rollout_kernels.hip, the optimizer in optim_kernels.hip.)
#include "mlp_launch.h"

using namespace mlp;

namespace {

inline char* align256(char* p) { return reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 255) & ~uintptr_t(255)); }

struct Carver {   // carves 256-byte aligned float arrays out of the caller's workspace
    char *p, *end;
    Carver(void* ws, size_t bytes) : p(align256((char*)ws)), end((char*)ws + bytes) {}
    float* take(size_t nfloat) {
        float* r = reinterpret_cast<float*>(p);
        p = align256(p + nfloat * sizeof(float));
        return r;
    }
    bool ok() const { return p <= end; }
};
inline size_t pad256(size_t nfloat) { return ((nfloat * sizeof(float) + 255) & ~size_t(255)) + 256; }

// y = (rew + shift) * scale + gamma * min(q1, q2)      (q2 == nullptr: q1 only)
__global__ void k_combine_target(int n, const float* __restrict__ rew, const float* __restrict__ q1,
                                 const float* __restrict__ q2, float shift, float scale, float gamma,
                                 float* __restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float q = q2 ? fminf(q1[i], q2[i]) : q1[i];
    y[i] = (rew[i] + shift) * scale + gamma * q;
}

// a += clip(sigma * eps, -c, c)      (td3.py:74-76)
__global__ void k_smooth(int n, float* __restrict__ a, const float* __restrict__ eps, float sigma, float c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    a[i] += fminf(fmaxf(sigma * eps[i], -c), c);
}

// the two above in one launch (mpg_td3_targets: the plain Q1 target of the priorities, then the smoothing of the action for the
// clipped double-Q target - two dependent ~5 us launches at any batch size)
__global__ void k_combine_and_smooth(int rows, int ad, const float* __restrict__ rew, const float* __restrict__ q1, float shift, float scale,
                                     float gamma, float* __restrict__ y1, float* __restrict__ a, const float* __restrict__ eps, float sigma,
                                     float c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows) y1[i] = (rew[i] + shift) * scale + gamma * q1[i];
    if (i < rows * ad) a[i] += fminf(fmaxf(sigma * eps[i], -c), c);
}

// y = sum_t gamma^t (r_t + shift) * scale + gamma^n q          (mpg_learner.py:165-168)
__global__ void k_nstep(int rows, int n, const float* __restrict__ rewards, const float* __restrict__ q, float shift,
                        float scale, float gamma, float* __restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    float acc = 0.f;
    for (int t = 0; t < n; ++t) acc += powf(gamma, (float)t) * ((rewards[(size_t)t * rows + i] + shift) * scale);
    y[i] = acc + powf(gamma, (float)n) * q[i];
}

// err = q - y; dz3 = err * inv_b; td (nullable) = err; loss_sum += 0.5 * inv_b * sum err^2 (one block, fixed order)
__global__ void __launch_bounds__(1024) k_q_err(int rows, const float* __restrict__ q, const float* __restrict__ y,
                                                float inv_b, float* __restrict__ dz3, float* __restrict__ td,
                                                float* __restrict__ loss_sum) {
    __shared__ float red[1024];
    float s = 0.f;
    for (int i = threadIdx.x; i < rows; i += 1024) {
        const float e = q[i] - y[i];
        dz3[i] = e * inv_b;
        if (td) td[i] = e;
        s += e * e;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_sum[0] = 0.5f * inv_b * red[0];
}

// TD3 policy loss pieces (td3.py:120-134): qmin = min(q1, q2); dL/dq_i = -inv_b where q_i is the smaller one
// (tf.reduce_min routes the gradient to the minimum; exact ties go to Q1).  One block, fixed-order sums.
__global__ void __launch_bounds__(1024) k_td3_dy(int rows, const float* __restrict__ q1, const float* __restrict__ q2,
                                                 float inv_b, float* __restrict__ dy1, float* __restrict__ dy2,
                                                 float* __restrict__ qmin_sum, float* __restrict__ qmin_sqsum) {
    __shared__ float red[2][1024];
    float s = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < rows; i += 1024) {
        const bool first = q1[i] <= q2[i];
        const float m = first ? q1[i] : q2[i];
        dy1[i] = first ? -inv_b : 0.f;
        dy2[i] = first ? 0.f : -inv_b;
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
        qmin_sum[0] = red[0][0];
        qmin_sqsum[0] = red[1][0];
    }
}

// Large batches (TD3 at B = 65 536: the one-block forms above walk 64 rows per thread, 18 - 34 us each): the same element-wise
// work over a grid, block b leaving the sums of its contiguous run of rows in part[b] (+ part[PARTS + b]); k_finish_parts adds the
// partials in block order.  Fixed order at both levels: deterministic.
constexpr int ERR_PARTS = 64;
__global__ void __launch_bounds__(1024) k_q_err_mb(int rows, const float* __restrict__ q, const float* __restrict__ y, float inv_b,
                                                   float* __restrict__ dz3, float* __restrict__ td, float* __restrict__ part) {
    __shared__ float red[1024];
    const int per = (rows + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
    float s = 0.f;
    for (int i = r0 + threadIdx.x; i < r1; i += 1024) {
        const float e = q[i] - y[i];
        dz3[i] = e * inv_b;
        if (td) td[i] = e;
        s += e * e;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void __launch_bounds__(1024) k_td3_dy_mb(int rows, const float* __restrict__ q1, const float* __restrict__ q2, float inv_b,
                                                    float* __restrict__ dy1, float* __restrict__ dy2, float* __restrict__ part) {
    __shared__ float red[2][1024];
    const int per = (rows + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
    float s = 0.f, s2 = 0.f;
    for (int i = r0 + threadIdx.x; i < r1; i += 1024) {
        const bool first = q1[i] <= q2[i];
        const float m = first ? q1[i] : q2[i];
        dy1[i] = first ? -inv_b : 0.f;
        dy2[i] = first ? 0.f : -inv_b;
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
    if (threadIdx.x == 0) { part[blockIdx.x] = red[0][0]; part[ERR_PARTS + blockIdx.x] = red[1][0]; }
}
// out0 = scale0 * sum_b part[b]; out1 (nullable) = scale1 * sum_b part[PARTS + b]
__global__ void k_finish_parts(int n_part, const float* __restrict__ part, float scale0, float scale1, float* __restrict__ out0,
                               float* __restrict__ out1) {
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < n_part; ++b) s += part[b];
        out0[0] = scale0 * s;
    } else if (threadIdx.x == 64 && out1) {
        float s = 0.f;
        for (int b = 0; b < n_part; ++b) s += part[ERR_PARTS + b];
        out1[0] = scale1 * s;
    }
}
constexpr int ERR_MB_MIN_ROWS = 8192;          // below: the one-block forms (one launch instead of two)

// out[i] ~ N(0, 1): Philox4x32-10(key = seed, counter = (i / 4, ctr)), Box-Muller on two of the four words per pair of outputs
__global__ void k_normal_fill(int n, uint32_t k0, uint32_t k1, uint32_t c1, uint32_t c2, float* __restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;          // one Philox draw = four normals
    if (4 * q >= n) return;
    const Philox4 p = philox4x32_10((uint32_t)q, c1, c2, 0x6e6f726du, k0, k1);
    const float r0 = sqrtf(-2.f * logf(u01(p.v[0]))), r1 = sqrtf(-2.f * logf(u01(p.v[2])));
    float s0, c0, s1, cc1;
    sincosf(6.283185307179586f * u01(p.v[1]), &s0, &c0);
    sincosf(6.283185307179586f * u01(p.v[3]), &s1, &cc1);
    const float z[4] = {r0 * c0, r0 * s0, r1 * cc1, r1 * s1};
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (4 * q + e < n) out[4 * q + e] = z[e];
}

// the priorities' td error of TD3 (td3.py:83-92): y1 - Q1(s, a) = (y1 - y) - (Q1(s, a) - y), the last term being the `td` output of
// the critic-loss pass
__global__ void k_td3_priority(int n, const float* __restrict__ y1, const float* __restrict__ y, const float* __restrict__ td,
                               float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (y1[i] - y[i]) - td[i];
}

// ga[row][k] = dx1[row][od + k] + dx2[row][od + k]
__global__ void k_sum_action_grad(int rows, int od, int ad, const float* __restrict__ dx1, const float* __restrict__ dx2,
                                  float* __restrict__ ga) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * ad) return;
    const int row = i / ad, k = i % ad;
    ga[i] = dx1[(long)row * (od + ad) + od + k] + dx2[(long)row * (od + ad) + od + k];
}

inline OutSpec policy_out(const mpg_cfg_t* c) {
    OutSpec o;
    const bool ranged = c->action_range > 0.f;
    o.out_tanh = (c->policy_out_act == MPG_ACT_TANH || ranged) ? 1 : 0;
    o.out_scale = ranged ? c->action_range : 1.f;
    o.sigma = 0.f; o.seed = 0; o.ctr = 0;
    return o;
}
inline OutSpec linear_out() {
    OutSpec o;
    o.out_tanh = 0; o.out_scale = 1.f; o.sigma = 0.f; o.seed = 0; o.ctr = 0;
    return o;
}

inline bool cfg_ok(const mpg_cfg_t* c) {
    // policy_out_activation='tanh' WITH an action_range would be range*tanh(tanh(z)) in the reference (policy.py:176-177,
    // 197-199); the kernels implement range*tanh(z) / tanh(z) / z only, so that combination is refused, not approximated
    return c && ((c->obs_dim >= 6 && c->obs_dim <= 16 && c->act_dim == 2) || (c->obs_dim == 4 && c->act_dim == 1)) &&
           !(c->policy_out_act == MPG_ACT_TANH && c->action_range > 0.f);
}

}  // namespace

extern "C" int mpg_mlp_forward(const float* params, int in_dim, int out_dim, int out_used, int out_act, int rows,
                               const float* x, const float* in_scale, int n_scaled, float* y, const mpg_wcache_t* wcache,
                               mpg_stream_t stream) {
    MPG_REQUIRE(params && x && y && rows > 0, "mpg_mlp_forward: null pointer / rows");
    OutSpec o = linear_out();
    o.out_tanh = out_act == MPG_ACT_TANH;
    mpg_cfg_t handles = {};                    // only the handle fields are read by the launcher
    handles.wcache[0] = wcache;
    const mpg_cfg_t* cfg = &handles;
    return launch_forward(cfg, params, in_dim, out_dim, out_used, rows, xspec(x, in_dim, nullptr, 0, in_scale, n_scaled), o, y,
                          out_used, nullptr, nullptr, mpg_stream(stream));
}

extern "C" int mpg_policy_action(const mpg_cfg_t* cfg, const float* policy_params, int rows, const float* obs,
                                 float explore_sigma, uint64_t seed, uint64_t ctr, float* act, mpg_stream_t stream) {
    MPG_REQUIRE(cfg_ok(cfg) && policy_params && obs && act && rows > 0, "mpg_policy_action: bad argument");
    OutSpec o = policy_out(cfg);
    o.sigma = explore_sigma; o.seed = seed; o.ctr = ctr;
    return launch_forward(cfg, policy_params, cfg->obs_dim, 2 * cfg->act_dim, cfg->act_dim, rows,
                          xspec(obs, cfg->obs_dim, nullptr, 0, cfg->obs_scale, cfg->obs_dim), o, act, cfg->act_dim,
                          nullptr, nullptr, mpg_stream(stream));
}

extern "C" size_t mpg_q_targets_workspace_bytes(const mpg_cfg_t* cfg, int rows) {
    if (!cfg_ok(cfg) || rows <= 0) return 0;
    return pad256((size_t)rows * cfg->act_dim) + 2 * pad256(rows);
}

extern "C" int mpg_q_targets(const mpg_cfg_t* cfg, const float* policy_t, const float* q1t, const float* q2t, int rows,
                             const float* rew, const float* obs_tp1, const float* smooth_eps, float smooth_sigma,
                             float smooth_clip, float* y, void* ws, size_t ws_bytes, mpg_stream_t stream) {
    MPG_REQUIRE(cfg_ok(cfg) && policy_t && q1t && rew && obs_tp1 && y && ws && rows > 0, "mpg_q_targets: bad argument");
    if (ws_bytes < mpg_q_targets_workspace_bytes(cfg, rows)) {
        mpg_set_error("mpg_q_targets: workspace too small");
        return MPG_EWORKSPACE;
    }
    hipStream_t s = mpg_stream(stream);
    Carver cv(ws, ws_bytes);
    float* a = cv.take((size_t)rows * cfg->act_dim);
    float* q1 = cv.take(rows);
    float* q2 = cv.take(rows);
    const int od = cfg->obs_dim, ad = cfg->act_dim;
    int rc = launch_forward(cfg, policy_t, od, 2 * ad, ad, rows, xspec(obs_tp1, od, nullptr, 0, cfg->obs_scale, od),
                            policy_out(cfg), a, ad, nullptr, nullptr, s);
    if (rc) return rc;
    if (smooth_eps) {
        const int n = rows * ad;
        hipLaunchKernelGGL(k_smooth, dim3((n + 255) / 256), dim3(256), 0, s, n, a, smooth_eps, smooth_sigma, smooth_clip);
        MPG_CHECK_LAUNCH("k_smooth");
    }
    const XSpec xq = xspec(obs_tp1, od, a, ad, cfg->obs_scale, od);
    rc = launch_forward(cfg, q1t, od + ad, 1, 1, rows, xq, linear_out(), q1, 1, nullptr, nullptr, s);
    if (rc) return rc;
    if (q2t) {
        rc = launch_forward(cfg, q2t, od + ad, 1, 1, rows, xq, linear_out(), q2, 1, nullptr, nullptr, s);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_combine_target, dim3((rows + 255) / 256), dim3(256), 0, s, rows, rew, q1, q2t ? q2 : nullptr,
                       cfg->rew_shift, cfg->rew_scale, cfg->gamma, y);
    MPG_CHECK_LAUNCH("k_combine_target");
    return MPG_OK;
}

// TD3 with prioritized replay needs TWO targets per minibatch: the clipped double-Q target with target-policy smoothing
// (td3.py:69-81) and the plain Q1 target of the priorities' td error (td3.py:83-92).  Both start from pi_t(s~'): evaluated ONCE
// here (the two mpg_q_targets calls of the method path evaluate it twice - one 65 536-row network pass, ~40 us, of every step).
extern "C" int mpg_td3_targets(const mpg_cfg_t* cfg, const float* policy_t, const float* q1t, const float* q2t, int rows,
                               const float* rew, const float* obs_tp1, const float* smooth_eps, float smooth_sigma,
                               float smooth_clip, float* y, float* y1, void* ws, size_t ws_bytes, mpg_stream_t stream) {
    MPG_REQUIRE(cfg_ok(cfg) && policy_t && q1t && q2t && rew && obs_tp1 && y && y1 && ws && rows > 0, "mpg_td3_targets: bad argument");
    if (ws_bytes < mpg_q_targets_workspace_bytes(cfg, rows)) {
        mpg_set_error("mpg_td3_targets: workspace too small");
        return MPG_EWORKSPACE;
    }
    hipStream_t s = mpg_stream(stream);
    Carver cv(ws, ws_bytes);
    float* a = cv.take((size_t)rows * cfg->act_dim);
    float* q1 = cv.take(rows);
    float* q2 = cv.take(rows);
    const int od = cfg->obs_dim, ad = cfg->act_dim;
    int rc = launch_forward(cfg, policy_t, od, 2 * ad, ad, rows, xspec(obs_tp1, od, nullptr, 0, cfg->obs_scale, od),
                            policy_out(cfg), a, ad, nullptr, nullptr, s);
    if (rc) return rc;
    const XSpec xq = xspec(obs_tp1, od, a, ad, cfg->obs_scale, od);
    rc = launch_forward(cfg, q1t, od + ad, 1, 1, rows, xq, linear_out(), q1, 1, nullptr, nullptr, s);      // Q1t(s~', pi_t(s~')): y1
    if (rc) return rc;
    if (smooth_eps) {
        const int n = rows * ad;
        hipLaunchKernelGGL(k_combine_and_smooth, dim3((n + 255) / 256), dim3(256), 0, s, rows, ad, rew, q1, cfg->rew_shift, cfg->rew_scale,
                           cfg->gamma, y1, a, smooth_eps, smooth_sigma, smooth_clip);
        MPG_CHECK_LAUNCH("k_combine_and_smooth");
        rc = launch_forward(cfg, q1t, od + ad, 1, 1, rows, xq, linear_out(), q1, 1, nullptr, nullptr, s);
        if (rc) return rc;
    } else {
        hipLaunchKernelGGL(k_combine_target, dim3((rows + 255) / 256), dim3(256), 0, s, rows, rew, q1, (const float*)nullptr, cfg->rew_shift,
                           cfg->rew_scale, cfg->gamma, y1);
        MPG_CHECK_LAUNCH("k_combine_target");
    }
    rc = launch_forward(cfg, q2t, od + ad, 1, 1, rows, xq, linear_out(), q2, 1, nullptr, nullptr, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_combine_target, dim3((rows + 255) / 256), dim3(256), 0, s, rows, rew, q1, q2, cfg->rew_shift, cfg->rew_scale,
                       cfg->gamma, y);
    MPG_CHECK_LAUNCH("k_combine_target");
    return MPG_OK;
}

extern "C" int mpg_normal_fill(int n, uint64_t seed, uint64_t ctr, float* out, mpg_stream_t stream) {
    MPG_REQUIRE(out && n > 0, "mpg_normal_fill: bad argument");
    hipLaunchKernelGGL(k_normal_fill, dim3(((n + 3) / 4 + 255) / 256), dim3(256), 0, mpg_stream(stream), n, (uint32_t)seed,
                       (uint32_t)(seed >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32), out);
    MPG_CHECK_LAUNCH("k_normal_fill");
    return MPG_OK;
}

extern "C" int mpg_td3_priority_errors(int rows, const float* y1, const float* y, const float* td, float* out, mpg_stream_t stream) {
    MPG_REQUIRE(y1 && y && td && out && rows > 0, "mpg_td3_priority_errors: bad argument");
    hipLaunchKernelGGL(k_td3_priority, dim3((rows + 255) / 256), dim3(256), 0, mpg_stream(stream), rows, y1, y, td, out);
    MPG_CHECK_LAUNCH("k_td3_priority");
    return MPG_OK;
}

extern "C" int mpg_nstep_targets(const mpg_cfg_t* cfg, const float* policy_t, const float* q1t, int rows, int n,
                                 const float* rewards, const float* last_obs, float* y, void* ws, size_t ws_bytes,
                                 mpg_stream_t stream) {
    MPG_REQUIRE(cfg_ok(cfg) && policy_t && q1t && rewards && last_obs && y && ws && rows > 0 && n > 0,
                "mpg_nstep_targets: bad argument");
    if (ws_bytes < mpg_q_targets_workspace_bytes(cfg, rows)) {
        mpg_set_error("mpg_nstep_targets: workspace too small");
        return MPG_EWORKSPACE;
    }
    hipStream_t s = mpg_stream(stream);
    Carver cv(ws, ws_bytes);
    float* a = cv.take((size_t)rows * cfg->act_dim);
    float* q1 = cv.take(rows);
    const int od = cfg->obs_dim, ad = cfg->act_dim;
    int rc = launch_forward(cfg, policy_t, od, 2 * ad, ad, rows, xspec(last_obs, od, nullptr, 0, cfg->obs_scale, od),
                            policy_out(cfg), a, ad, nullptr, nullptr, s);
    if (rc) return rc;
    rc = launch_forward(cfg, q1t, od + ad, 1, 1, rows, xspec(last_obs, od, a, ad, cfg->obs_scale, od), linear_out(), q1, 1,
                        nullptr, nullptr, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_nstep, dim3((rows + 255) / 256), dim3(256), 0, s, rows, n, rewards, q1, cfg->rew_shift,
                       cfg->rew_scale, cfg->gamma, y);
    MPG_CHECK_LAUNCH("k_nstep");
    return MPG_OK;
}

extern "C" size_t mpg_q_loss_grad_workspace_bytes(const mpg_cfg_t* cfg, int rows) {
    if (!cfg_ok(cfg) || rows <= 0) return 0;
    const int in = cfg->obs_dim + cfg->act_dim;
    return 4 * pad256(stash_floats(rows)) + 2 * pad256(rows) + pad256(wgrad_workspace_floats(rows, in, 1)) + pad256(2 * ERR_PARTS);
}

extern "C" int mpg_q_loss_grad(const mpg_cfg_t* cfg, const float* q_params, int rows, const float* obs, const float* act,
                               const float* y, float inv_b_global, float* loss_sum, float* grad, float* td, void* ws,
                               size_t ws_bytes, mpg_stream_t stream) {
    MPG_REQUIRE(cfg_ok(cfg) && q_params && obs && act && y && loss_sum && grad && ws && rows > 0,
                "mpg_q_loss_grad: bad argument");
    if (ws_bytes < mpg_q_loss_grad_workspace_bytes(cfg, rows)) {
        mpg_set_error("mpg_q_loss_grad: workspace too small (%zu < %zu)", ws_bytes, mpg_q_loss_grad_workspace_bytes(cfg, rows));
        return MPG_EWORKSPACE;
    }
    hipStream_t s = mpg_stream(stream);
    const int od = cfg->obs_dim, ad = cfg->act_dim, in = od + ad;
    Carver cv(ws, ws_bytes);
    float* h1 = cv.take(stash_floats(rows));
    float* h2 = cv.take(stash_floats(rows));
    float* dz1 = cv.take(stash_floats(rows));
    float* dz2 = cv.take(stash_floats(rows));
    float* q = cv.take(rows);
    float* dz3 = cv.take(rows);
    float* slabs = cv.take(wgrad_workspace_floats(rows, in, 1));
    float* parts = cv.take(2 * ERR_PARTS);
    const XSpec xq = xspec(obs, od, act, ad, cfg->obs_scale, od);
    int rc = launch_forward(cfg, q_params, in, 1, 1, rows, xq, linear_out(), q, 1, h1, h2, s);
    if (rc) return rc;
    // the thin parameter gradients ride in the backward launch (mlp_launch.h): its per-workgroup partials live where the dz1 stash
    // would (never larger), the weight-gradient launch reads h1 and dz2 only
    const bool thin = backward_takes_thin(in, 1);
    // large batches: the loss partials of the 64-block error kernel are added up by one extra block of the gradient's summation launch
    // (round 5; k_finish_parts' arithmetic) when that launch exists (thin), by k_finish_parts otherwise
    FinishJob fin{parts, ERR_PARTS, ERR_PARTS, 0.5f * inv_b_global, 0.f, loss_sum, nullptr};
    const bool mb = rows >= ERR_MB_MIN_ROWS;
    if (mb) {
        hipLaunchKernelGGL(k_q_err_mb, dim3(ERR_PARTS), dim3(1024), 0, s, rows, q, y, inv_b_global, dz3, td, parts);
        if (!thin) hipLaunchKernelGGL(k_finish_parts, dim3(1), dim3(128), 0, s, ERR_PARTS, parts, 0.5f * inv_b_global, 0.f, loss_sum, (float*)nullptr);
    } else
    hipLaunchKernelGGL(k_q_err, dim3(1), dim3(1024), 0, s, rows, q, y, inv_b_global, dz3, td, loss_sum);
    MPG_CHECK_LAUNCH("k_q_err");
    rc = launch_backward(cfg, q_params, in, 1, 1, rows, dz3, 1, nullptr, 0, 0, 1.f, h1, h2, thin ? nullptr : dz1, dz2, nullptr, nullptr, 0, s,
                         thin ? &xq : nullptr, thin ? dz1 : nullptr);
    if (rc) return rc;
    return launch_wgrad(cfg, in, 1, 1, rows, xq, h1, h2, dz1, dz2, dz3, inv_b_global, grad, slabs, s, thin, thin ? dz1 : nullptr,
                        thin ? backward_thin_parts(rows) : 0, (mb && thin) ? &fin : nullptr);
}

extern "C" size_t mpg_td3_policy_grad_workspace_bytes(const mpg_cfg_t* cfg, int rows) {
    if (!cfg_ok(cfg) || rows <= 0) return 0;
    const int od = cfg->obs_dim, ad = cfg->act_dim, qin = od + ad;
    return 8 * pad256(stash_floats(rows)) + 2 * pad256((size_t)rows * ad) + 4 * pad256(rows) +
           2 * pad256((size_t)rows * qin) + pad256((size_t)rows * ad) + pad256(wgrad_workspace_floats(rows, od, 2 * ad)) +
           pad256(2 * ERR_PARTS);
}

extern "C" int mpg_td3_policy_grad(const mpg_cfg_t* cfg, const float* policy_params, const float* q1, const float* q2,
                                   int rows, const float* obs, float inv_b_global, float* qmin_sum, float* qmin_sqsum,
                                   float* grad, void* ws, size_t ws_bytes, mpg_stream_t stream) {
    MPG_REQUIRE(cfg_ok(cfg) && policy_params && q1 && q2 && obs && qmin_sum && qmin_sqsum && grad && ws && rows > 0,
                "mpg_td3_policy_grad: bad argument");
    if (ws_bytes < mpg_td3_policy_grad_workspace_bytes(cfg, rows)) {
        mpg_set_error("mpg_td3_policy_grad: workspace too small");
        return MPG_EWORKSPACE;
    }
    hipStream_t s = mpg_stream(stream);
    const int od = cfg->obs_dim, ad = cfg->act_dim, qin = od + ad;
    Carver cv(ws, ws_bytes);
    float* hp1 = cv.take(stash_floats(rows)); float* hp2 = cv.take(stash_floats(rows));
    float* h11 = cv.take(stash_floats(rows)); float* h12 = cv.take(stash_floats(rows));
    float* h21 = cv.take(stash_floats(rows)); float* h22 = cv.take(stash_floats(rows));
    float* dz1 = cv.take(stash_floats(rows)); float* dz2 = cv.take(stash_floats(rows));
    float* a = cv.take((size_t)rows * ad); float* dz3 = cv.take((size_t)rows * ad);
    float* qv1 = cv.take(rows); float* qv2 = cv.take(rows); float* dy1 = cv.take(rows); float* dy2 = cv.take(rows);
    float* dx1 = cv.take((size_t)rows * qin); float* dx2 = cv.take((size_t)rows * qin);
    float* ga = cv.take((size_t)rows * ad);
    float* slabs = cv.take(wgrad_workspace_floats(rows, od, 2 * ad));
    float* parts = cv.take(2 * ERR_PARTS);
    const OutSpec po = policy_out(cfg);
    const XSpec xp = xspec(obs, od, nullptr, 0, cfg->obs_scale, od);
    int rc = launch_forward(cfg, policy_params, od, 2 * ad, ad, rows, xp, po, a, ad, hp1, hp2, s);        // td3.py:123
    if (rc) return rc;
    const XSpec xq = xspec(obs, od, a, ad, cfg->obs_scale, od);
    rc = launch_forward(cfg, q1, qin, 1, 1, rows, xq, linear_out(), qv1, 1, h11, h12, s);                  // :124
    if (rc) return rc;
    rc = launch_forward(cfg, q2, qin, 1, 1, rows, xq, linear_out(), qv2, 1, h21, h22, s);                  // :125
    if (rc) return rc;
    const bool thin = backward_takes_thin(od, ad);        // (see mpg_q_loss_grad)
    FinishJob fin{parts, ERR_PARTS, ERR_PARTS, 1.f, 1.f, qmin_sum, qmin_sqsum};
    const bool mb = rows >= ERR_MB_MIN_ROWS;
    if (mb) {
        hipLaunchKernelGGL(k_td3_dy_mb, dim3(ERR_PARTS), dim3(1024), 0, s, rows, qv1, qv2, inv_b_global, dy1, dy2, parts);
        if (!thin) hipLaunchKernelGGL(k_finish_parts, dim3(1), dim3(128), 0, s, ERR_PARTS, parts, 1.f, 1.f, qmin_sum, qmin_sqsum);
    } else
    hipLaunchKernelGGL(k_td3_dy, dim3(1), dim3(1024), 0, s, rows, qv1, qv2, inv_b_global, dy1, dy2, qmin_sum, qmin_sqsum);
    MPG_CHECK_LAUNCH("k_td3_dy");
    rc = launch_backward(cfg, q1, qin, 1, 1, rows, dy1, 1, nullptr, 0, 0, 1.f, h11, h12, nullptr, nullptr, nullptr, dx1, qin, s);
    if (rc) return rc;
    rc = launch_backward(cfg, q2, qin, 1, 1, rows, dy2, 1, nullptr, 0, 0, 1.f, h21, h22, nullptr, nullptr, nullptr, dx2, qin, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sum_action_grad, dim3((rows * ad + 255) / 256), dim3(256), 0, s, rows, od, ad, dx1, dx2, ga);
    MPG_CHECK_LAUNCH("k_sum_action_grad");
    rc = launch_backward(cfg, policy_params, od, 2 * ad, ad, rows, ga, ad, a, ad, po.out_tanh, po.out_scale, hp1, hp2, thin ? nullptr : dz1,
                         dz2, dz3, nullptr, 0, s, thin ? &xp : nullptr, thin ? dz1 : nullptr);
    if (rc) return rc;
    return launch_wgrad(cfg, od, 2 * ad, ad, rows, xp, hp1, hp2, dz1, dz2, dz3, inv_b_global, grad, slabs, s, thin, thin ? dz1 : nullptr,
                        thin ? backward_thin_parts(rows) : 0, (mb && thin) ? &fin : nullptr);
}
