// Host-side launchers of the generic network kernels (internal to libmpg_hip.so; not part of the C ABI).
#pragma once
#include "mlp_core.h"

namespace mlp {

struct XSpec {              // network input = [x0 (d0 columns, column i < n_scaled multiplied by scale[i]) | x1 (d1 columns)]
    const float* x0;
    int d0, ld0;            // ld = row stride in floats
    const float* x1;
    int d1, ld1;
    float scale[24];        // 1 beyond n_scaled (n_scaled <= 16: mpg_cfg_t.obs_scale has 16 entries)
    int n_scaled;
};

inline XSpec xspec(const float* x0, int d0, const float* x1, int d1, const float* scale, int n_scaled) {
    XSpec s;
    s.x0 = x0; s.d0 = d0; s.ld0 = d0; s.x1 = x1; s.d1 = d1; s.ld1 = d1; s.n_scaled = n_scaled;
    for (int i = 0; i < 24; ++i) s.scale[i] = (scale && i < n_scaled && i < 16) ? scale[i] : 1.f;
    return s;
}

// one partial of the thin parameter gradients = the network's flat layout without W2: W1 [in][H] | b1 | b2 | W3 [H][out] | b3
__host__ __device__ inline int thin_floats(int in_dim, int out_dim) { return in_dim * H + H + H + H * out_dim + out_dim; }

#ifdef __HIPCC__
// loads one row group of the network input into sX [16][xs_of<IN>()] (zero padded; width = d0 + d1 columns)
template <int IN>
__device__ __forceinline__ void load_x_group(const XSpec& x, int rows, long g, float* sX) {
    constexpr int XSW = xs_of<IN>();
    const int tid = threadIdx.x;
    if (tid < GROUP * XSW) {
        const int row = tid / XSW, i = tid % XSW;
        const long gr = g * GROUP + row;
        float v = 0.f;
        if (gr < rows && i < x.d0 + x.d1) {
            if (i < x.d0)
                v = x.x0[gr * x.ld0 + i] * x.scale[i];
            else
                v = x.x1[gr * x.ld1 + (i - x.d0)];
        }
        sX[tid] = v;
    }
}

#endif

struct OutSpec {            // y = out_scale * tanh(z) (out_tanh) or z; optional Philox N(0, sigma) exploration noise
    int out_tanh;
    float out_scale;
    float sigma;
    uint64_t seed, ctr;
};

// y[rows][ldy] (first `ou` columns) = net(x); optional G16 stashes of both hidden activations.
// cfg (nullable): source of the caller's weight-cache descriptors and kernel timer, nothing else is read from it here
int launch_forward(const mpg_cfg_t* cfg, const float* params, int in_dim, int out_dim, int ou, int rows, const XSpec& x,
                   const OutSpec& o, float* y, int ldy, float* h1, float* h2, hipStream_t s);

// Input-side backward: dy [rows][lddy] is dL/d(output after activation); yout [rows][ldyo] the forward outputs.
// Writes (each nullable) dz1, dz2 (G16), dz3 [rows][ou] and dx [rows][lddx] (in_dim columns, w.r.t. the network
// input as seen by the first layer, i.e. after scaling).
int launch_backward(const mpg_cfg_t* cfg, const float* params, int in_dim, int out_dim, int ou, int rows, const float* dy, int lddy,
                    const float* yout, int ldyo, int out_tanh, float out_scale, const float* h1, const float* h2,
                    float* dz1, float* dz2, float* dz3, float* dx, int lddx, hipStream_t s, const XSpec* thin_x = nullptr,
                    float* thin_part = nullptr);
// thin_part (with thin_x = the network input, dx == nullptr, backward_takes_thin): the thin parameter gradients (dW1, db1, db2, dW3,
// db3) are accumulated by the backward launch itself: one partial of thin_floats(in_dim, out_dim) floats per workgroup,
// backward_thin_parts(rows) of them (never more than the dz1 stash of the same rows holds: callers alias it).  dz1 is then NOT
// written; follow with launch_wgrad(no_thin = true) and launch_thin_reduce.
bool backward_takes_thin(int in_dim, int ou);
int backward_thin_parts(int rows);
int launch_thin_reduce(const float* part, int n_part, int in_dim, int out_dim, float* grad, hipStream_t s);

// Weight gradient of one network over `rows` rows from stashes; result (net_size floats, fully reduced over rows,
// accumulate == 0: overwritten) in grad.  ws must hold wgrad_workspace_floats(rows, in_dim, out_dim) floats.
size_t wgrad_workspace_floats(int rows, int in_dim, int out_dim);
// inv_b: the scale the upstream gradients carry (1/B_global, / M for tiled rollouts); informational - the fp16 split operands are
// centred per chunk from the data itself (mlp_wgrad.h)
// no_thin: only dW2 is computed (the thin parts of `grad` are left ZERO: the caller adds them - rollout_common.h thin_floats)
// a pending scalar reduction that rides in the weight-gradient launch's summation kernel (one block more) instead of a launch of its
// own: out0 = scale0 * sum_b part[b], out1 (nullable) = scale1 * sum_b part[stride1 + b], b < n_part, in block order
struct FinishJob {
    const float* part;
    int n_part, stride1;
    float scale0, scale1;
    float *out0, *out1;
};
int launch_wgrad(const mpg_cfg_t* cfg, int in_dim, int out_dim, int ou, int rows, const XSpec& x, const float* h1, const float* h2,
                 const float* dz1, const float* dz2, const float* dz3, float inv_b, float* grad, float* ws, hipStream_t s,
                 bool no_thin = false, const float* thin_part = nullptr, int n_thin_part = 0, const struct FinishJob* fin = nullptr);
// (no_thin with thin_part: the thin partials of the backward launch / reverse sweep are summed in the same launch as the chunk slabs -
// no launch_thin_reduce afterwards)

// ---- fused critic-side kernels (one 16-row group per workgroup; callers fall back to the unfused launchers when a
// configuration is not covered) ----------------------------------------------------------------------------------
struct NetRef {            // one network: Keras-ordered parameters (+ optional packed images resolved by the launcher)
    const float* params;
    int in_dim, out_dim;
};

// y = (rew + shift) * scale + gamma * min_i Qt_i(s~', a'),  a' = pi_t(s~') (+ clipped smoothing noise): ONE launch.
// draw (nullable): the minibatch is drawn from the replay ring inside the same launch and written to draw_out
struct DrawOut {
    float *obs, *act, *rew, *obs2;
};
int launch_target_fused(const mpg_cfg_t* cfg, const float* policy_t, const float* q1t, const float* q2t, int rows,
                        const float* rew, const float* obs_tp1, const float* smooth_eps, float sigma, float clipc, float* y,
                        hipStream_t s, const mpg_replay_draw_t* draw = nullptr, const DrawOut* draw_out = nullptr,
                        float* qpart = nullptr);

struct CriticStash {       // G16 stashes + dz3 of one critic, kept for the weight-gradient launch
    float *h1, *h2, *dz1, *dz2, *dz3;
};
// forward + error + input-side backward of n_q (1 or 2) critics in ONE launch.  loss_part [n_q][ngroups] receives
// 0.5*inv_b*sum(err^2) per row group; td (nullable) = Q1(s~,a) - y.
int launch_qloss_fused(const mpg_cfg_t* cfg, const float* const* q_params, int n_q, int rows, const float* obs,
                       const float* act, const float* y, float inv_b, const CriticStash* st, float* loss_part, float* td,
                       hipStream_t s);

// critic value + input gradient at the selected rollout slices in ONE launch: q = Q(xq), per-group partial sums of
// the returns G + gpow*q and of their squares (ret_part [n_sel][ngroups][2]), dx = coef_k * dQ/dxq.  xq [n_sel*R][qin].
int launch_qslice_fused(const mpg_cfg_t* cfg, const float* q_params, int qin, int R, int n_sel, const float* xq, const float* gk, const float* gpow,
                        const float* coef, float* ret_part, float* gxq, hipStream_t s);

// weight gradients of up to 3 networks in one launch + one reduce launch.  Also folds the fixed-order sums of the
// per-group loss / return partials (sum_src[i] has sum_n[i] floats -> sum_dst[i]), up to 8 of them.
struct WgradJob {
    int in_dim, out_dim, ou, rows;
    XSpec x;
    const float *h1, *h2, *dz1, *dz2, *dz3;
    float inv_b;           // scale carried by dz (see launch_wgrad)
    float* grad;
    float* slabs;          // wgrad_workspace_floats(rows, in_dim, out_dim)
};
struct SumJob {
    const float* src;
    int n, stride;         // sums src[0], src[stride], ... (n terms)
    float* dst;
};
int launch_wgrad_multi(const mpg_cfg_t* cfg, const WgradJob* jobs, int n_jobs, const SumJob* sums, int n_sums, float* sq_part, hipStream_t s,
                       int phases = 3, int first_job = 0);
// critic losses + critic at the two selected slices in one launch (launch_qloss_fused + launch_qslice_fused with n_sel == 2)
int launch_critic_fused(const mpg_cfg_t* cfg, const float* const* q_params, int n_q, int rows, const float* obs,
                        const float* act, const float* y, float inv_b, const CriticStash* st, float* loss_part, const float* xq,
                        const float* gk, const float* gpow, const float* coef, float* ret_part, float* gxq, hipStream_t s,
                        const float* qpart = nullptr, const float* rew = nullptr, float* y_out = nullptr);

inline size_t stash_floats(int rows) { return (size_t)((rows + GROUP - 1) / GROUP) * GROUP * H; }

}  // namespace mlp
