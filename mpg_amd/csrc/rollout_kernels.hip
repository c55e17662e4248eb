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
#include <algorithm>
#include <stdlib.h>

#include "mlp_launch.h"

using namespace mlp;

namespace {

constexpr int MAXN = 32;      // horizon limit (reference default n = 25)
constexpr int MAXSEL = 4;     // slices entering the loss (reference default {0, 25})
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
// forward sweep
// ---------------------------------------------------------------------------------------------------------------
struct RollArgs {
    const float* policy;
    int rows, M, n;                     // R = rows * M trajectories, horizon n
    float obs_scale[8];
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
    float* dbg;                         // diagnostic builds only
};

template <class ENV>
__global__ void __launch_bounds__(NTHREAD, 2) k_rollout_fwd(const RollArgs a) {
    constexpr int OBS = ENV::OBS, ACT = ENV::ACT, QIN = OBS + ACT;
    __shared__ __attribute__((aligned(16))) float smem[GROUP * LDA + GROUP * XS + NWAVE * GROUP * MAXOUT + MAXN * GROUP];
    float* sA = smem;
    float* sX = sA + GROUP * LDA;
    float* sPart = sX + GROUP * XS;
    float* sEps = sPart + NWAVE * GROUP * MAXOUT;
    __shared__ float sGp[MAXN];
    const Lane L;
    const int tid = threadIdx.x;
    if (tid <= a.n) sGp[tid] = powf(a.gamma, (float)tid);     // tf.pow(gamma, ri) in float32, mpg_learner.py:245
    const Net net = make_net(a.policy, OBS, 2 * ACT);
    float w2[128];
    SmallRegs<OBS, ACT> r;
    if (a.pack) load_w2_packed(a.pack, L, w2); else load_w2_fwd(net.W2, L, w2);
    load_small<OBS, ACT>(net, L, r);
    float b3r[2] = {0.f, 0.f};                         // output bias in registers: no global load on the serial chain
#pragma unroll
    for (int k = 0; k < ACT; ++k) b3r[k] = net.b3[k];
    const long R = (long)a.rows * a.M;
    const long ngroups = (R + GROUP - 1) / GROUP;
#ifdef MPG_STAMP
    if ((tid & 63) == 0) {
        for (int k = 0; k < 10; ++k) g_st_acc[tid >> 6][k] = 0;
        g_st_prev[tid >> 6] = __builtin_amdgcn_s_memtime();
    }
#endif
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const long tr = g * GROUP + tid;               // this lane's trajectory (tid < 16 only)
        const bool own = tid < GROUP, live = own && tr < R;
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float G = 0.f;
        float act_first[2] = {0.f, 0.f};
        if (live) {
            const float* src = a.obs0 + (tr % a.rows) * OBS;
#pragma unroll
            for (int i = 0; i < OBS; ++i) o[i] = src[i];
            if (a.act0) {
#pragma unroll
                for (int k = 0; k < ACT; ++k) act_first[k] = a.act0[(tr % a.rows) * ACT + k];
            }
        }
        // the whole group's model noise goes to LDS up front (one value per thread), off the serial chain: either the
        // caller's eps or Philox draws.  Visible to the dynamics lanes after the first barrier of the step loop.
        for (int idx = tid; idx < a.n * GROUP; idx += NTHREAD) {
            const int t = idx / GROUP;
            const long trj = g * GROUP + (idx % GROUP);
            float z = 0.f;
            if (a.eps) {
                if (trj < R) z = a.eps[(long)t * R + trj];
            } else {
                const Philox4 p = philox4x32_10((uint32_t)trj, (uint32_t)t, a.nc0, a.nc1 ^ 0x6e6f6973u, a.nk0, a.nk1);
                z = sqrtf(-2.f * logf(u01(p.v[0]))) * cosf(6.283185307179586f * u01(p.v[1]));
            }
            sEps[idx] = z;
        }
        // Per step: B0 (input published) -> layer 1 -> barrier -> layer-2 MFMA block -> output partials -> B2 -> the
        // trajectory lanes' serial chain (tanh, action-dependent part of the model step, publish the next input).
        // Everything the chain does not strictly need sits in the trajectory wave's idle time before B2: it is the
        // older wave of its SIMD and leaves the MFMA block ~4000 cycles before the younger ones.
        float act[2] = {0.f, 0.f}, rew = 0.f;
        // record the action of step tb, its critic-input part and the discounted reward (late by one step: off the chain)
        auto book = [&](int tb) {
            if (live) {
                if (a.SA) {
                    float* rec = a.SA + ((long)tb * R + tr) * SAW + OBS;
#pragma unroll
                    for (int k = 0; k < ACT; ++k) rec[k] = act[k];
                }
                for (int ks = 0; ks < a.n_sel; ++ks)
                    if (a.sel[ks] == tb) {
                        float* xq = a.XQ + ((long)ks * R + tr) * QIN + OBS;
#pragma unroll
                        for (int k = 0; k < ACT; ++k) xq[k] = act[k];
                    }
            }
            if (tb < a.n) G += sGp[tb] * ((rew + a.rew_shift) * a.rew_scale);                 // mpg_learner.py:245
        };
        if (own) {
#pragma unroll
            for (int i = 0; i < XS; ++i) sX[tid * XS + i] = i < OBS ? o[i] * a.obs_scale[i] : 0.f;
        }
        for (int t = 0; t <= a.n; ++t) {
            lds_barrier();
            MPG_STAMP_AT(0);
            float h1[2][4], h2[2][4];
            forward_group<OBS, ACT, false>(sX, sA, sPart, L, w2, r, h1, h2, a.H1, (long)t * ngroups + g);
            if (a.H1) stash_store(a.H2, (long)t * ngroups + g, L, h2);
            float pre[ENV::NPRE];
            if (own) {
                if (t > 0) book(t - 1);
                if (live) {
                    if (a.SA) {
                        float* rec = a.SA + ((long)t * R + tr) * SAW;
#pragma unroll
                        for (int i = 0; i < OBS; ++i) rec[i] = o[i];
                    }
                    for (int ks = 0; ks < a.n_sel; ++ks)
                        if (a.sel[ks] == t) {
                            float* xq = a.XQ + ((long)ks * R + tr) * QIN;
#pragma unroll
                            for (int i = 0; i < OBS; ++i) xq[i] = o[i] * a.obs_scale[i];
                            a.GK[(long)ks * R + tr] = G;
                        }
                }
                if (t < a.n) ENV::pre(o, sEps[t * GROUP + tid], pre);
            }
            MPG_STAMP_AT(6);
            lds_barrier();
            MPG_STAMP_AT(5);
            if (own) {
#pragma unroll
                for (int k = 0; k < ACT; ++k) {
                    const float z = out_preact_tree(sPart, b3r[k], tid, k);
                    act[k] = a.out_tanh ? a.out_scale * fast_tanh(z) : z;
                }
                if (t == 0 && a.act0) {
#pragma unroll
                    for (int k = 0; k < ACT; ++k) act[k] = act_first[k];
                }
                if (t < a.n) {
                    float on[8];
                    ENV::finish(pre, act, on, rew);
#pragma unroll
                    for (int i = 0; i < XS; ++i) sX[tid * XS + i] = i < OBS ? on[i] * a.obs_scale[i] : 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = on[i];
                }
            }
            // sX of the next step is ordered behind this step's reads by the two barriers above
            MPG_STAMP_AT(7);
        }
        if (own) book(a.n);
#ifdef MPG_STAMP
        if ((tid & 63) == 0 && a.dbg)
            for (int k = 0; k < 8; ++k) a.dbg[((long)blockIdx.x * NWAVE + (tid >> 6)) * 8 + k] = (float)g_st_acc[tid >> 6][k];
#endif
    }
}

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

// ---------------------------------------------------------------------------------------------------------------
// reverse sweep
// ---------------------------------------------------------------------------------------------------------------
struct RollBwdArgs {
    const float* policy;
    int rows, M, n;
    float obs_scale[8];
    int out_tanh;
    float out_scale;
    const float *H1, *H2, *SA;
    int sel[MAXSEL], n_sel;
    const float* GXQ;                   // [n_sel][R][OBS+ACT] dL/d(critic input) at the selected slices
    float rho[MAXN];                    // dL/d(raw reward of step t)
    int stash_all;                      // 0: parameter gradient through step 0 only (MPG); 1: every step (NADP)
    float *DZ1, *DZ2, *DZ3;             // stashes for the weight gradient: T = stash_all ? n+1 : 1 steps
    const float* pack;                  // nullable: packed backward image of the policy's W2
    float* dbg;                         // diagnostic builds only
};

template <class ENV>
__global__ void __launch_bounds__(NTHREAD, 2) k_rollout_bwd(const RollBwdArgs a) {
    constexpr int OBS = ENV::OBS, ACT = ENV::ACT, QIN = OBS + ACT;
    __shared__ __attribute__((aligned(16))) float smem[2 * GROUP * LDA + GROUP * MAXOUT + NWAVE * GROUP * XS];
    float* sA = smem;
    float* sA1 = sA + GROUP * LDA;
    float* sD3 = sA1 + GROUP * LDA;
    float* sPartX = sD3 + GROUP * MAXOUT;
    // carry state of the 16 trajectory lanes between steps (adjoint of the next obs, record of the next step): kept in
    // LDS because registers are allocated for all 512 lanes while only 16 use them (the kernel sits at the 256 VGPR limit)
    __shared__ __attribute__((aligned(16))) float sCarry[GROUP * 16];
    const Lane L;
    const int tid = threadIdx.x;
    const Net net = make_net(a.policy, OBS, 2 * ACT);
    float w2t[128];
    SmallRegs<OBS, ACT> r;
    if (a.pack) load_w2_packed(a.pack, L, w2t); else load_w2_bwd(net.W2, L, w2t);
    load_small<OBS, ACT>(net, L, r);
    const long R = (long)a.rows * a.M;
    const long ngroups = (R + GROUP - 1) / GROUP;
#ifdef MPG_STAMP
    if ((tid & 63) == 0) {
        for (int k = 0; k < 10; ++k) g_st_acc[tid >> 6][k] = 0;
        g_st_prev[tid >> 6] = __builtin_amdgcn_s_memtime();
    }
#endif
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const long tr = g * GROUP + tid;
        const bool own = tid < GROUP, live = own && tr < R;
        if (tid < GROUP) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sCarry[tid * 16 + i] = 0.f;   // [0..8): dL/d(obs_{t+1}), [8..16): record t+1
        }
        float lam[8];
        // Software pipeline over the steps: the (obs | action) record and the h2 stash of step t-1 are requested while
        // step t computes, so that no HBM / L2 latency sits on the serial chain.
        float rec_cur[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float rec_pre[8];
        float h2_cur[2][4], h2_pre[2][4];
        if (live) {
            const f32x4* rp = reinterpret_cast<const f32x4*>(a.SA + ((long)a.n * R + tr) * SAW);
            const f32x4 r0 = rp[0], r1 = rp[1];
#pragma unroll
            for (int i = 0; i < 4; ++i) { rec_cur[i] = r0[i]; rec_cur[4 + i] = r1[i]; }
        }
        stash_load(a.H2, (long)a.n * ngroups + g, L, h2_cur);
        for (int t = a.n; t >= 0; --t) {
            float h1[2][4];
            if (own) {
                float ga[2] = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 8; ++i) lam[i] = 0.f;
                float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, act[2] = {0.f, 0.f};
                if (live) {
#pragma unroll
                    for (int i = 0; i < OBS; ++i) o[i] = rec_cur[i];
#pragma unroll
                    for (int k = 0; k < ACT; ++k) act[k] = rec_cur[OBS + k];
                    if (t < a.n) {
                        float on[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int i = 0; i < OBS; ++i) on[i] = sCarry[tid * 16 + 8 + i];
                        float lam_next[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) lam_next[i] = sCarry[tid * 16 + i];
                        ENV::vjp(o, act, on, lam_next, a.rho[t], lam, ga);
                    }
                    for (int ks = 0; ks < a.n_sel; ++ks)
                        if (a.sel[ks] == t) {
                            const float* gx = a.GXQ + ((long)ks * R + tr) * QIN;
#pragma unroll
                            for (int i = 0; i < OBS; ++i) lam[i] += gx[i] * a.obs_scale[i];
#pragma unroll
                            for (int k = 0; k < ACT; ++k) ga[k] += gx[OBS + k];
                        }
                }
#pragma unroll
                for (int k = 0; k < ACT; ++k) {
                    float d = ga[k];
                    if (a.out_tanh) {
                        const float th = act[k] / a.out_scale;
                        d *= a.out_scale * (1.f - th * th);
                    }
                    sD3[d3_index(tid, k)] = d;
                    if (live && a.DZ3 && (a.stash_all || t == 0))
                        a.DZ3[((long)(a.stash_all ? t : 0) * R + tr) * ACT + k] = d;
                }
            }
            float dz1[2][4], dz2[2][4];
            lds_barrier();
            MPG_STAMP_AT(0);
            backward_dz2<OBS, ACT>(sD3, sA, L, r, h2_cur, dz2);
            // all global loads of the step are issued HERE, behind the dz2 phase: h1 is consumed after the MFMA block,
            // the record and h2 stash of step t-1 in the next iteration (software pipeline)
            stash_load(a.H1, (long)t * ngroups + g, L, h1);
            if (t > 0) {
                if (live) {
                    const f32x4* rp = reinterpret_cast<const f32x4*>(a.SA + ((long)(t - 1) * R + tr) * SAW);
                    const f32x4 r0 = rp[0], r1 = rp[1];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { rec_pre[i] = r0[i]; rec_pre[4 + i] = r1[i]; }
                }
                stash_load(a.H2, (long)(t - 1) * ngroups + g, L, h2_pre);
            }
            if (t > 0)
                backward_rest<OBS, ACT, true>(sA, sA1, sPartX, L, w2t, r, h1, dz1);
            else
                backward_rest<OBS, ACT, false>(sA, sA1, sPartX, L, w2t, r, h1, dz1);
            if (a.DZ1 && (a.stash_all || t == 0)) {
                const long sg = (long)(a.stash_all ? t : 0) * ngroups + g;
                stash_store(a.DZ1, sg, L, dz1);
                stash_store(a.DZ2, sg, L, dz2);
            }
            if (own) {
                if (t > 0) {
                    float dxr[XS];
                    dx_reduce_row(sPartX, tid, dxr);
#pragma unroll
                    for (int i = 0; i < OBS; ++i) lam[i] += dxr[i] * a.obs_scale[i];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    sCarry[tid * 16 + i] = lam[i];
                    sCarry[tid * 16 + 8 + i] = rec_cur[i];
                    rec_cur[i] = rec_pre[i];
                }
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int j = 0; j < 4; ++j) h2_cur[tt][j] = h2_pre[tt][j];
            MPG_STAMP_AT(7);
            // next iteration: sD3 is rewritten by wave 0 only after it has passed backward_group's final barrier,
            // and read by the others only after the __syncthreads above -> no extra barrier needed.
        }
#ifdef MPG_STAMP
        if ((tid & 63) == 0 && a.dbg)
            for (int k = 0; k < 8; ++k) a.dbg[((long)blockIdx.x * NWAVE + (tid >> 6)) * 8 + k] = (float)g_st_acc[tid >> 6][k];
#endif
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
    return c && ((c->obs_dim == 6 && c->act_dim == 2 && c->env_kind == MPG_ENV_PATH_TRACKING) ||
                 (c->obs_dim == 4 && c->act_dim == 1 && c->env_kind == MPG_ENV_INVERTED_PENDULUM));
}

void fill_roll(RollArgs& a, const mpg_cfg_t* cfg, const float* policy, int rows, int M, int n) {
    a.policy = policy; a.rows = rows; a.M = M; a.n = n;
    for (int i = 0; i < 8; ++i) a.obs_scale[i] = i < cfg->obs_dim ? cfg->obs_scale[i] : 1.f;
    a.rew_scale = cfg->rew_scale; a.rew_shift = cfg->rew_shift; a.gamma = cfg->gamma;
    const bool ranged = cfg->action_range > 0.f;
    a.out_tanh = (cfg->policy_out_act == MPG_ACT_TANH || ranged) ? 1 : 0;
    a.out_scale = ranged ? cfg->action_range : 1.f;
    a.pack = weight_cache_lookup(make_net(policy, cfg->obs_dim, 2 * cfg->act_dim).W2, 0);
}

int grid_for(long ngroups) { return (int)(ngroups < 256 ? ngroups : 256); }

struct PgLayout {
    size_t h, sa, xq, gk, q, dyq, hq, gxq, dz, dz3, slabs, small, total;
};

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
    l.total = 2 * pad256(l.h) + pad256(l.sa) + pad256(l.xq) + 3 * pad256(l.gk) + 2 * pad256(l.hq) + pad256(l.gxq) +
              2 * pad256(l.dz) + pad256(l.dz3) + pad256(l.slabs) + 2 * pad256(l.small);
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
    fa.dbg = nullptr;
#ifdef MPG_STAMP
    static float* s_dbg = nullptr;
    static int s_calls = 0;
    if (!s_dbg) (void)hipMalloc(&s_dbg, 256 * 8 * 8 * sizeof(float));
    fa.dbg = s_dbg;
#endif
    mpg_prof_begin(0, s);
    if (cfg->env_kind == MPG_ENV_PATH_TRACKING)
        hipLaunchKernelGGL((k_rollout_fwd<PathTracking>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, fa);
    else
        hipLaunchKernelGGL((k_rollout_fwd<Pendulum>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, fa);
    mpg_prof_end(0, s);
    MPG_CHECK_LAUNCH("k_rollout_fwd");
#ifdef MPG_STAMP
    if (++s_calls % 50 == 0) {
        static float h[256 * 8 * 8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, s_dbg, sizeof(h), hipMemcpyDeviceToHost);
        const int nwg = grid_for(ngroups);
        for (int w = 0; w < 8; ++w) {
            double acc[8] = {0};
            for (int b = 0; b < nwg; ++b) for (int k = 0; k < 8; ++k) acc[k] += h[(b * 8 + w) * 8 + k];
            fprintf(stderr, "[stamp fwd] wave %d cycles/step:", w);
            double tot = 0;
            for (int k = 0; k < 8; ++k) { fprintf(stderr, " p%d=%.0f", k, acc[k] / nwg / (n + 1)); tot += acc[k] / nwg / (n + 1); }
            fprintf(stderr, " total=%.0f\n", tot);
        }
    }
#endif
    return MPG_OK;
}

int run_rollout_bwd(const mpg_cfg_t* cfg, const float* policy_params, int rows, int M, int n, const int* select, int n_select,
                    const float* rho, const float* H1, const float* H2, const float* SA, const float* GXQ,
                    int all_steps_param_grad, float* DZ1, float* DZ2, float* DZ3, hipStream_t s) {
    const long R = (long)rows * M;
    const long ngroups = (R + GROUP - 1) / GROUP;
    const int od = cfg->obs_dim, ad = cfg->act_dim;
    RollArgs fa;
    fill_roll(fa, cfg, policy_params, rows, M, n);
    for (int k = 0; k < MAXSEL; ++k) fa.sel[k] = k < n_select ? select[k] : -1;
    // ---- reverse sweep ----
    RollBwdArgs ba;
    ba.policy = policy_params; ba.rows = rows; ba.M = M; ba.n = n;
    for (int i = 0; i < 8; ++i) ba.obs_scale[i] = fa.obs_scale[i];
    ba.out_tanh = fa.out_tanh; ba.out_scale = fa.out_scale;
    ba.H1 = H1; ba.H2 = H2; ba.SA = SA; ba.n_sel = n_select;
    for (int k = 0; k < MAXSEL; ++k) ba.sel[k] = fa.sel[k];
    ba.GXQ = GXQ;
    for (int t = 0; t < MAXN; ++t) ba.rho[t] = rho[t];
    ba.stash_all = all_steps_param_grad ? 1 : 0;
    ba.DZ1 = DZ1; ba.DZ2 = DZ2; ba.DZ3 = DZ3;
    ba.pack = weight_cache_lookup(make_net(policy_params, od, 2 * ad).W2, 1);
    ba.dbg = nullptr;
#ifdef MPG_STAMP
    static float* s_dbg_b = nullptr;
    static int s_calls_b = 0;
    if (!s_dbg_b) (void)hipMalloc(&s_dbg_b, 256 * 8 * 8 * sizeof(float));
    ba.dbg = s_dbg_b;
#endif
    mpg_prof_begin(1, s);
    if (cfg->env_kind == MPG_ENV_PATH_TRACKING)
        hipLaunchKernelGGL((k_rollout_bwd<PathTracking>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, ba);
    else
        hipLaunchKernelGGL((k_rollout_bwd<Pendulum>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, ba);
    mpg_prof_end(1, s);
    MPG_CHECK_LAUNCH("k_rollout_bwd");
#ifdef MPG_STAMP
    if (++s_calls_b % 50 == 0) {
        static float h[256 * 8 * 8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, s_dbg_b, sizeof(h), hipMemcpyDeviceToHost);
        const int nwg = grid_for(ngroups);
        for (int w = 0; w < 8; ++w) {
            double acc[8] = {0};
            for (int b = 0; b < nwg; ++b) for (int k = 0; k < 8; ++k) acc[k] += h[(b * 8 + w) * 8 + k];
            fprintf(stderr, "[stamp bwd] wave %d cycles/step:", w);
            double tot = 0;
            for (int k = 0; k < 8; ++k) { fprintf(stderr, " p%d=%.0f", k, acc[k] / nwg / (n + 1)); tot += acc[k] / nwg / (n + 1); }
            fprintf(stderr, " total=%.0f\n", tot);
        }
    }
#endif
    return MPG_OK;
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

    // ---- forward sweep ----
    int rc = run_rollout_fwd(cfg, policy_params, rows, M, n, select, n_select, obs0, eps, noise_seed, noise_ctr, H1, H2, SA, XQ,
                             GK, s);
    if (rc) return rc;
    const long ngroups = (R + GROUP - 1) / GROUP;
    (void)ngroups;

    // ---- critic at the selected slices: values, returns, input gradients ----
    OutSpec lin; lin.out_tanh = 0; lin.out_scale = 1.f; lin.sigma = 0.f; lin.seed = lin.ctr = 0;
    const int RQ = (int)(n_select * R);
    rc = launch_forward(q1_params, qin, 1, 1, RQ, xspec(XQ, qin, nullptr, 0, nullptr, 0), lin, Q, 1, HQ1, HQ2, s);
    if (rc) return rc;
    const Coefs cf = make_coefs(cfg, select, n_select, w, inv_b_global, M);
    RetCoef rcf;
    for (int k = 0; k < MAXSEL; ++k) { rcf.gpow[k] = cf.gpow[k]; rcf.coef[k] = cf.coef[k]; }
    hipLaunchKernelGGL(k_returns, dim3(1), dim3(1024), 0, s, rows, M, n_select, rcf, Q, GK, DYQ, ret_sum, ret_sqsum);
    MPG_CHECK_LAUNCH("k_returns");
    rc = launch_backward(q1_params, qin, 1, 1, RQ, DYQ, 1, nullptr, 0, 0, 1.f, HQ1, HQ2, nullptr, nullptr, nullptr, GXQ, qin, s);
    if (rc) return rc;

    // ---- reverse sweep ----
    rc = run_rollout_bwd(cfg, policy_params, rows, M, n, select, n_select, cf.rho, H1, H2, SA, GXQ, all_steps_param_grad, DZ1, DZ2,
                         DZ3, s);
    if (rc) return rc;

    // ---- policy weight gradient from the stashes (step 0 only, or every step for NADP) ----
    const int T = all_steps_param_grad ? n + 1 : 1;
    XSpec xs = xspec(SA, od, nullptr, 0, cfg->obs_scale, od);
    xs.ld0 = SAW;
    return launch_wgrad(od, 2 * ad, ad, (int)(T * R), xs, H1, H2, DZ1, DZ2, DZ3, grad, slabs, s);
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
    if (cfg->env_kind == MPG_ENV_PATH_TRACKING)
        hipLaunchKernelGGL((k_rollout_fwd<PathTracking>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, fa);
    else
        hipLaunchKernelGGL((k_rollout_fwd<Pendulum>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, fa);
    MPG_CHECK_LAUNCH("k_rollout_fwd (q target)");
    OutSpec lin; lin.out_tanh = 0; lin.out_scale = 1.f; lin.sigma = 0.f; lin.seed = lin.ctr = 0;
    int rc = launch_forward(q1t, qin, 1, 1, rows, xspec(XQ, qin, nullptr, 0, nullptr, 0), lin, Q, 1, nullptr, nullptr, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_gq, dim3((rows + 255) / 256), dim3(256), 0, s, rows, GK, Q, powf(cfg->gamma, (float)n), y);
    MPG_CHECK_LAUNCH("k_gq");
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
                    2 * pad256(stash_floats(R)) + pad256(l.dz3p) + pad256(l.slab_p);
    l.fallback0 = std::max(mpg_q_targets_workspace_bytes(cfg, rows), mpg_q_loss_grad_workspace_bytes(cfg, rows));
    l.fallback1 = mpg_rollout_pg_workspace_bytes(cfg, rows, M, n, n_sel, 0);
    return l;
}

}  // namespace

extern "C" size_t mpg_mpg_gradients_workspace_bytes(const mpg_cfg_t* cfg, int rows, int M, int n, int n_select, int n_q) {
    if (!cfg_ok(cfg) || rows <= 0 || M <= 0 || n <= 0 || n >= MAXN || n_select <= 0 || n_select > MAXSEL || n_q < 1 || n_q > 2)
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
    const bool fused = rows % GROUP == 0 && M == 1;
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

    const float* y = y_in;
    if (!y) {   // 1. clipped double-Q (or single-Q) target, mpg_learner.py:126-134
        const DrawOut dout{obs, act, rew, obs_tp1};
        int rc = launch_target_fused(cfg, policy_t, qt[0], qt[1], rows, rew, obs_tp1, nullptr, 0.f, 0.f, y_out, s, draw, &dout);
        if (rc) return rc;
        y = y_out;
    }
    int rc;
    const Coefs cf = make_coefs(cfg, select, n_select, w, inv_b_global, 1);
    static const bool merged_critic = getenv("MPG_NO_CRITIC_FUSED") == nullptr;    // A/B switch (tools only)
    if (n_select == 2 && merged_critic) {
        // 2. rollout forward sweep
        rc = run_rollout_fwd(cfg, policy, rows, 1, n, select, n_select, obs, eps, noise_seed, noise_ctr, H1, H2, SA, XQ, GK, s);
        if (rc) return rc;
        // 3.+4. critics (forward, error, input-side backward, mpg_learner.py:326-354) and the critic at the two selected
        //       slices (returns and input gradients) in one launch
        rc = launch_critic_fused(cfg, qp, n_q, rows, obs, act, y, inv_b_global, st, loss_part, XQ, GK, cf.gpow, cf.coef, ret_part,
                                 GXQ, s);
        if (rc) return rc;
    } else {
        // 2. critics: forward, error, input-side backward (mpg_learner.py:326-354)
        rc = launch_qloss_fused(cfg, qp, n_q, rows, obs, act, y, inv_b_global, st, loss_part, nullptr, s);
        if (rc) return rc;
        // 3. rollout forward sweep
        rc = run_rollout_fwd(cfg, policy, rows, 1, n, select, n_select, obs, eps, noise_seed, noise_ctr, H1, H2, SA, XQ, GK, s);
        if (rc) return rc;
        // 4. critic at the selected slices: returns and input gradients
        rc = launch_qslice_fused(qp[0], qin, rows, n_select, XQ, GK, cf.gpow, cf.coef, ret_part, GXQ, s);
        if (rc) return rc;
    }
    // 5. reverse sweep
    rc = run_rollout_bwd(cfg, policy, rows, 1, n, select, n_select, cf.rho, H1, H2, SA, GXQ, 0, DZ1, DZ2, DZ3, s);
    if (rc) return rc;
    // 6./7. weight gradients of every network + all scalar statistics, two launches
    WgradJob jobs[3];
    const XSpec xq = xspec(obs, od, act, ad, cfg->obs_scale, od);
    for (int k = 0; k < n_q; ++k) {
        jobs[k].in_dim = qin; jobs[k].out_dim = 1; jobs[k].ou = 1; jobs[k].rows = rows; jobs[k].x = xq;
        jobs[k].h1 = st[k].h1; jobs[k].h2 = st[k].h2; jobs[k].dz1 = st[k].dz1; jobs[k].dz2 = st[k].dz2; jobs[k].dz3 = st[k].dz3;
        jobs[k].grad = gq[k]; jobs[k].slabs = slab_q[k];
    }
    WgradJob& jp = jobs[n_q];
    jp.in_dim = od; jp.out_dim = 2 * ad; jp.ou = ad; jp.rows = rows;
    jp.x = xspec(SA, od, nullptr, 0, cfg->obs_scale, od);
    jp.x.ld0 = SAW;
    jp.h1 = H1; jp.h2 = H2; jp.dz1 = DZ1; jp.dz2 = DZ2; jp.dz3 = DZ3; jp.grad = gp; jp.slabs = slab_p;
    const int ngroups = rows / GROUP;
    SumJob sums[8];
    int ns = 0;
    for (int k = 0; k < n_q; ++k) { sums[ns].src = loss_part + (size_t)k * ngroups; sums[ns].n = ngroups; sums[ns].stride = 1; sums[ns].dst = stats + k; ++ns; }
    for (int k = 0; k < n_select && ns + 1 < 8; ++k) {
        sums[ns].src = ret_part + (size_t)k * ngroups * 2; sums[ns].n = ngroups; sums[ns].stride = 2; sums[ns].dst = stats + 2 + k; ++ns;
        sums[ns].src = ret_part + (size_t)k * ngroups * 2 + 1; sums[ns].n = ngroups; sums[ns].stride = 2; sums[ns].dst = stats + 2 + n_select + k; ++ns;
    }
    MPG_REQUIRE(n_q + 2 * n_select <= 8, "mpg_mpg_gradients: too many statistics (n_select <= 3 with two critics)");
    return launch_wgrad_multi(jobs, n_q + 1, sums, ns, sq_part, s);
}
