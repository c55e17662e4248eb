// Shared by the three translation units of the fused rollout (rollout_fwd.hip, rollout_bwd.hip, rollout_kernels.hip):
// the differentiable models, the kernel argument blocks and the launch entry points.  The two sweeps live in their own
// translation units so that each can be compiled with the scheduling options that suit it (mpg_amd/build.py).
#pragma once
#include <algorithm>
#include <stdlib.h>

#include "mlp_launch.h"

namespace rollout {

using namespace mlp;


constexpr int MAXN = 32;      // horizon limit (reference default n = 25)
constexpr int MAXSEL = 4;     // slices entering the loss (reference default {0, 25})
constexpr int MAXF = MPG_ENV_MAX_FUTURE;   // look-ahead entries of an observation (path_tracking_env.py:385-402)
constexpr int SAW = 8;        // floats per (step, trajectory) record: obs | action

// ---------------------------------------------------------------------------------------------------------------
// differentiable models: one lane = one trajectory
// ---------------------------------------------------------------------------------------------------------------
// The model step sits on the serial chain of the rollout (16 lanes work, 496 wait), so it uses the hardware
// reciprocal / exp / sin / cos (<= 1-2 ulp, far inside the stated 1e-4 gradient tolerance) instead of the
// correctly-rounded library routines.
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ void fast_sincos(float x, float* s, float* c) {   // |x| <= pi here
    *s = __sinf(x);
    *c = __cosf(x);
}
__device__ __forceinline__ float fast_tanh(float z) {                       // 1 - 2/(e^{2z} + 1); exact limits +-1
    return 1.f - 2.f * frcp(__expf(2.f * z) + 1.f);
}

// The forward step is split so that the part that needs the LATE-arriving action is a handful of fmas:
//   pre(obs, eps) -> everything that does not depend on the action (the new state is affine in the action);
//   finish(pre, action) -> new obs, raw reward.
// pre runs before the barrier that publishes the output-layer partials, so the serial chain between two policy
// evaluations only contains tanh + finish.  (The same split of the adjoint - Jacobian entries ahead of time, a
// matrix-vector product on the chain - was tried for the reverse sweep: it needs ~30 more live registers at a point
// where the kernel has none, the compiler spilled part of the stationary weights, 158 -> 218 us.)
struct PathTracking {
    static constexpr int OBS = 6, ACT = 2;
    static constexpr int NPRE = 9;
    static constexpr int FUT_SRC = 3;     // PathTrackingModel._get_obs (path_tracking_env.py:262-268): every look-ahead entry of a MODEL
                                          // observation is a copy of delta_y (entry 3); only the start observation carries real ones
    // vehicle parameters, path_tracking_env.py:60-68; tau = 1/10 (:248)
    static constexpr float C_f = -128915.5f, C_r = -85943.6f, A = 1.06f, B = 1.85f, MASS = 1412.f, I_z = 1536.7f;
    static constexpr float TAU = 0.1f;
    static constexpr float K1 = TAU * (A * C_f - B * C_r), K2 = TAU * C_f, K3 = TAU * MASS, K4 = TAU * (C_f + C_r);
    static constexpr float K5 = TAU * A * C_f, K6 = TAU * (A * A * C_f + B * B * C_r);
    static constexpr float S0 = (float)(1.2 * 3.14159265358979323846 / 9.0), S1 = 3.f;   // action scaling :282
    static constexpr float PI_F = 3.14159265358979323846f;

    // obs -> veh state is a shift of entry 0 by 20 (:268-277); we carry obs and add the shift on use.
    // One model step (f_xu :78-138 with tau = 0.1, rewards :181-199 on the PRE-step state and scaled action).
    // eps: standard normal (noise = 0.5 + 0.01 eps, :119).
    // p: [0] nvx without the action term, [1],[2] nvy = p1 + p2*de, [3],[4] nr = p3 + p4*de, [5] ndy, [6] ndphi,
    //    [7] nx, [8] state part of -reward
    __device__ static void pre(const float (&o)[8], float eps, float (&p)[NPRE]) {
        const float vx = o[0] + 20.f, vy = o[1], r = o[2], dy = o[3], dphi = o[4], x = o[5];
        const float iD1 = frcp(MASS * vx - K4), iD2 = frcp(K6 - I_z * vx);
        p[0] = vx + TAU * (vy * r);
        p[1] = (MASS * vy * vx + K1 * r - K3 * vx * vx * r) * iD1;
        p[2] = -K2 * vx * iD1;
        p[3] = (-I_z * r * vx - K1 * vy) * iD2;
        p[4] = K5 * vx * iD2;
        float sp, cp;
        fast_sincos(dphi, &sp, &cp);
        p[5] = dy + TAU * (vx * sp + vy * cp) + (0.5f + 0.01f * eps);
        float ndphi = dphi + TAU * r;
        if (ndphi > PI_F) ndphi -= 2.f * PI_F;                   // :290
        if (ndphi <= -PI_F) ndphi += 2.f * PI_F;                 // :291
        p[6] = ndphi;
        p[7] = x + TAU * (vx * cp - vy * sp);
        const float dv = vx - 20.f;
        p[8] = 0.01f * dv * dv + 0.04f * dy * dy + 0.1f * dphi * dphi + 0.02f * r * r;
    }
    __device__ static void finish(const float (&p)[NPRE], const float (&a)[2], float (&on)[8], float& rew) {
        const float de = a[0] * S0, ax = a[1] * S1;
        const float nvx = fminf(fmaxf(fmaf(TAU, ax, p[0]), 1.f), 35.f);      // :289
        on[0] = nvx - 20.f;
        on[1] = fmaf(p[2], de, p[1]);
        on[2] = fmaf(p[4], de, p[3]);
        on[3] = p[5]; on[4] = p[6]; on[5] = p[7]; on[6] = 0.f; on[7] = 0.f;
        rew = -(p[8] + 5.f * de * de + 0.05f * ax * ax);
    }
    __device__ static void step(const float (&o)[8], const float (&a)[2], float eps, float (&on)[8], float& rew) {
        float p[NPRE];
        pre(o, eps, p);
        finish(p, a, on, rew);
    }

    // adjoint of step(): lam = dL/d(new obs), rho = dL/d(raw reward).  Returns dL/d(obs) and dL/d(action).
    // (oracle/mpg_oracle.py:pt_model_step_vjp is the float64 statement of the same formulas)
    __device__ static void vjp(const float (&o)[8], const float (&a)[2], const float (&onext)[8], const float (&lam)[8],
                               float rho, float (&g)[8], float (&ga)[2]) {
        const float vx = o[0] + 20.f, vy = o[1], r = o[2], dy = o[3], dphi = o[4];
        const float de = a[0] * S0, ax = a[1] * S1;
        const float nvx_raw = vx + TAU * (ax + vy * r);
        const float l_vx = (nvx_raw >= 1.f && nvx_raw <= 35.f) ? lam[0] : 0.f;
        const float l_vy = lam[1], l_r = lam[2], l_dy = lam[3], l_dphi = lam[4], l_x = lam[5];
        const float D1 = MASS * vx - K4, D2 = K6 - I_z * vx;
        const float iD1 = frcp(D1), iD2 = frcp(D2);
        const float nvy = (MASS * vy * vx + K1 * r - K2 * de * vx - K3 * vx * vx * r) * iD1;
        const float nr = (-I_z * r * vx - K1 * vy + K5 * de * vx) * iD2;
        float sp, cp;
        fast_sincos(dphi, &sp, &cp);
        const float dvy_vx = (MASS * vy - K2 * de - 2.f * K3 * vx * r - nvy * MASS) * iD1;
        const float dvy_vy = MASS * vx * iD1;
        const float dvy_r = (K1 - K3 * vx * vx) * iD1;
        const float dvy_de = -K2 * vx * iD1;
        const float dr_vx = (-I_z * r + K5 * de + nr * I_z) * iD2;
        const float dr_vy = -K1 * iD2;
        const float dr_r = -I_z * vx * iD2;
        const float dr_de = K5 * vx * iD2;
        g[0] = l_vx + l_vy * dvy_vx + l_r * dr_vx + l_dy * TAU * sp + l_x * TAU * cp + rho * (-0.02f * (vx - 20.f));
        g[1] = l_vx * TAU * r + l_vy * dvy_vy + l_r * dr_vy + l_dy * TAU * cp - l_x * TAU * sp;
        g[2] = l_vx * TAU * vy + l_vy * dvy_r + l_r * dr_r + l_dphi * TAU + rho * (-0.04f * r);
        g[3] = l_dy + rho * (-0.08f * dy);
        g[4] = l_dphi + l_dy * TAU * (vx * cp - vy * sp) - l_x * TAU * (vx * sp + vy * cp) + rho * (-0.2f * dphi);
        g[5] = l_x;
        ga[0] = (l_vy * dvy_de + l_r * dr_de + rho * (-10.f * de)) * S0;
        ga[1] = (l_vx * TAU + rho * (-0.1f * ax)) * S1;
        (void)onext;
    }
};

struct Pendulum {
    static constexpr int OBS = 4, ACT = 1;
    static constexpr int NPRE = 6;
    static constexpr int FUT_SRC = 0;     // (no look-ahead entries)
    // inverted_pendulum_model.py:18-26,38-44: m = 9.42, m1 = 4.89, m2 = 0, l1 = 0.6
    static constexpr float D1c = 9.42f + 4.89f, D2c = 0.5f * 4.89f * 0.6f, D4c = (1.f / 3.f) * 4.89f * 0.6f * 0.6f;
    static constexpr float F1c = 0.5f * 4.89f * 0.6f * 9.81f, TAU = 0.04f;

    // p: [0] new p, [1] new theta, [2],[3] new pdot = p2 + p3*a, [4],[5] new thetadot = p4 + p5*a
    __device__ static void pre(const float (&o)[8], float eps, float (&p)[NPRE]) {
        const float pos = o[0], th = o[1], pd = o[2], thd = o[3];
        float sn, c;
        sincosf(th, &sn, &c);                                               // theta is not range-limited: keep the library routine
        const float idet = frcp(D1c * D4c - D2c * D2c * c * c);             // closed-form 2x2 inverse (:53)
        const float F1s = D2c * sn * thd * thd, F2 = F1c * sn;              // F1 = F1s + u, u = 100 a (action_trans :96-97)
        p[0] = pos + TAU * pd + (0.1f + 0.5f * eps);                        // :57,:61
        p[1] = th + TAU * thd;
        p[2] = pd + TAU * ((D4c * F1s - D2c * c * F2) * idet);
        p[3] = TAU * 100.f * D4c * idet;
        p[4] = thd + TAU * ((-D2c * c * F1s + D1c * F2) * idet);
        p[5] = -TAU * 100.f * D2c * c * idet;
    }
    __device__ static void finish(const float (&p)[NPRE], const float (&a)[2], float (&on)[8], float& rew) {
        on[0] = p[0];
        on[1] = p[1];
        on[2] = fmaf(p[3], a[0], p[2]);
        on[3] = fmaf(p[5], a[0], p[4]);
        on[4] = on[5] = on[6] = on[7] = 0.f;
        rew = -(0.01f * on[0] * on[0] + on[1] * on[1]) - (1e-3f * on[2] * on[2] + 1e-3f * on[3] * on[3]);   // :66-73,:93
    }
    __device__ static void step(const float (&o)[8], const float (&a)[2], float eps, float (&on)[8], float& rew) {
        float p[NPRE];
        pre(o, eps, p);
        finish(p, a, on, rew);
    }

    __device__ static void vjp(const float (&o)[8], const float (&a)[2], const float (&onext)[8], const float (&lam_in)[8],
                               float rho, float (&g)[8], float (&ga)[2]) {
        // the reward is taken on the NEW (noisy) state: fold it into the adjoint of the new state first
        const float l_p = lam_in[0] + rho * (-0.02f * onext[0]);
        const float l_th = lam_in[1] + rho * (-2.f * onext[1]);
        const float l_pd = lam_in[2] + rho * (-2e-3f * onext[2]);
        const float l_thd = lam_in[3] + rho * (-2e-3f * onext[3]);
        const float th = o[1], thd = o[3];
        const float u = 100.f * a[0];
        float sn, c;
        sincosf(th, &sn, &c);
        const float det = D1c * D4c - D2c * D2c * c * c, idet = frcp(det);
        const float F1 = D2c * sn * thd * thd + u, F2 = F1c * sn;
        const float pdd = (D4c * F1 - D2c * c * F2) * idet;
        const float thdd = (-D2c * c * F1 + D1c * F2) * idet;
        const float ddet_th = 2.f * D2c * D2c * c * sn;
        const float dF1_th = D2c * c * thd * thd, dF1_thd = 2.f * D2c * sn * thd, dF2_th = F1c * c;
        const float dpdd_th = (D4c * dF1_th + D2c * sn * F2 - D2c * c * dF2_th - pdd * ddet_th) * idet;
        const float dthdd_th = (D2c * sn * F1 - D2c * c * dF1_th + D1c * dF2_th - thdd * ddet_th) * idet;
        const float dpdd_thd = D4c * dF1_thd * idet, dthdd_thd = -D2c * c * dF1_thd * idet;
        const float dpdd_u = D4c * idet, dthdd_u = -D2c * c * idet;
        g[0] = l_p;
        g[1] = l_th + TAU * (l_pd * dpdd_th + l_thd * dthdd_th);
        g[2] = l_p * TAU + l_pd;
        g[3] = l_th * TAU + l_thd + TAU * (l_pd * dpdd_thd + l_thd * dthdd_thd);
        ga[0] = 100.f * TAU * (l_pd * dpdd_u + l_thd * dthdd_u);
        ga[1] = 0.f;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// kernel argument blocks
// ---------------------------------------------------------------------------------------------------------------
struct RollArgs {
    const float* policy;
    int rows, M, n;                     // R = rows * M trajectories, horizon n
    int obs_dim;                        // ENV::OBS + number of look-ahead entries (PathTracking: 6 .. 14)
    float obs_scale[16];
    float rew_scale, rew_shift, gamma;
    int out_tanh;
    float out_scale;
    const float* obs0;                  // [rows][OBS]
    const float* act0;                  // nullable [rows][ACT]: first action given (NADP Q-target rollout)
    const float* eps;                   // [n][R] standard normal, or nullptr: Philox4x32-10(noise_seed, noise_ctr, t, trajectory)
    uint32_t nk0, nk1, nc0, nc1;
    float *H1, *H2;                     // nullable G16 stashes, group index t*ngroups + g
    float* SA;                          // nullable [(n+1)][R][SAW]: obs | action of every step
    int sel[MAXSEL], n_sel;
    float* XQ;                          // [n_sel][R][OBS+ACT] critic inputs (scaled obs | action) at the selected slices
    float* GK;                          // [n_sel][R] discounted reward sums G_k
    const float* pack;                  // nullable: packed forward image of the policy's W2
    int* status;                        // nullable: MPG_STATUS_* word of the caller
    float* dbg;                         // diagnostic builds only
};

struct RollBwdArgs {
    const float* policy;
    int rows, M, n;
    int obs_dim;
    float obs_scale[16];
    int out_tanh;
    float out_scale, inv_out_scale;     // (the reciprocal from the host: a division on the serial chain is ten instructions)
    const float *H1, *H2, *SA;
    int sel[MAXSEL], n_sel;
    const float* GXQ;                   // [n_sel][R][OBS+ACT] dL/d(critic input) at the selected slices
    float rho[MAXN];                    // dL/d(raw reward of step t)
    int stash_all;                      // 0: parameter gradient through step 0 only (MPG); 1: every step (NADP)
    float *DZ1, *DZ2, *DZ3;             // stashes for the weight gradient: T = stash_all ? n+1 : 1 steps
    const float* pack;                  // nullable: packed backward image of the policy's W2
    float* thin_part;                   // THIN instantiations only: [gridDim.x][thin_floats(OBS, 2 ACT)] per-workgroup sums (below)
    float* dbg;                         // diagnostic builds only
};

// "Thin" parameter gradients accumulated INSIDE the reverse sweep (round 4; NADP, nadp.py:128-194: the policy's parameter gradient
// flows through all n + 1 evaluations).  dW1, db1, db2, dW3, db3 need dz1, dz2, h2, dz3 and the network input of every (step,
// row) - all of which the reverse sweep holds in registers / LDS at the moment it produces them.  Left to the weight-gradient
// launch they are two more 218 MB stashes to write here and to read there (config 3: k_wgrad<4,1> moved 872 MB in 250 us, the
// largest kernel of the step).  With THIN the sweep keeps per-lane running sums in LDS (the other seven waves do this while
// wave 0 runs the serial chain), leaves one partial vector per workgroup and skips the dz1 stash; the weight-gradient launch then
// reads only h1 and dz2 for dW2 and launch_thin_reduce adds the partials.  Layout of a partial = the network's flat layout without W2:
// [W1 (in x 256) | b1 | b2 | W3 (256 x out) | b3].
// (thin_floats: mlp_launch.h)

inline void fill_roll(RollArgs& a, const mpg_cfg_t* cfg, const float* policy, int rows, int M, int n) {
    a.policy = policy; a.rows = rows; a.M = M; a.n = n;
    a.obs_dim = cfg->obs_dim;
    for (int i = 0; i < 16; ++i) a.obs_scale[i] = i < cfg->obs_dim ? cfg->obs_scale[i] : 1.f;
    a.rew_scale = cfg->rew_scale; a.rew_shift = cfg->rew_shift; a.gamma = cfg->gamma;
    const bool ranged = cfg->action_range > 0.f;
    a.out_tanh = (cfg->policy_out_act == MPG_ACT_TANH || ranged) ? 1 : 0;
    a.out_scale = ranged ? cfg->action_range : 1.f;
    a.pack = weight_cache_lookup(cfg, make_net(policy, cfg->obs_dim, 2 * cfg->act_dim).W2, 0);
    a.status = mpg_status_of(cfg);
}

inline int grid_for(long ngroups) { return (int)(ngroups < 256 ? ngroups : 256); }

// launch of the forward / reverse sweep kernel for the environment `env_kind` (n: horizon, only for the diagnostics)
// prof (nullable): the caller's kernel timer
int launch_rollout_fwd(const RollArgs& a, int env_kind, long ngroups, int n, hipStream_t s, mpg_prof_t* prof);
int launch_rollout_bwd(const RollBwdArgs& a, int env_kind, long ngroups, int n, hipStream_t s, mpg_prof_t* prof);

}  // namespace rollout
