// K1: PathTrackingEnv - vectorised real-environment stepping for gfx950.
//
// One lane per agent; the 20 sub-steps of VehicleDynamics.simulation run in registers, the agent's state
// block is read once and written once (SoA [8][n], coalesced).  The arithmetic follows the reference
// operation-by-operation (envs_and_models/path_tracking_env.py:78-138,144-179,181-199,204-220,410-487):
// this translation unit is compiled with -ffp-contract=off so that every float32 product and sum rounds
// exactly where numpy / TF round it; only the transcendental library calls (sin, cos, atan, tan, log,
// sqrt) differ from the reference's by their last-ulp behaviour.
//
// State block (opaque to callers): v_x, v_y, r, y, phi, x, delta_y, delta_phi.
// Observations carry 6 + K entries, K = num_future_data look-ahead delta-y terms (path_tracking_env.py:385-402).
// env_kind MPG_ENV_INVERTED_PENDULUM is dispatched to env_cart_pole.hip.
#include "env_internal.h"

// The worker's policy pass rides in the env launch (k_policy_step_store_reset, below).  The network engine is compiled here as
// every other translation unit compiles it - contraction allowed - although this file is built with -ffp-contract=off for the
// environment's op-by-op arithmetic: the pragma is lexical, the environment code below is back under "off".
#pragma clang fp contract(fast)
#include "mlp_core.h"

namespace worker_policy {
using namespace mlp;

struct Args {
    const float* params;       // policy network, Keras order
    const float* pack;         // nullable: packed forward image of W2 (weight cache)
    int* status;               // nullable: MPG_STATUS_* word of the caller
    int out_tanh;
    float out_scale, sigma;
    uint32_t k0, k1, c1, c2;   // exploration-noise key and counter
    float scale[8];            // obs_scale (1 beyond obs_dim)
};

constexpr int SMEM_FLOATS = A_IMG + GROUP * 8 + NWAVE * GROUP * MAXOUT;

// One 16-row group of mpg_policy_action: k_forward<6, 2, PK, 1> (mlp_kernels.hip) for unit g - the same device functions on the
// same operands in the same order, so the actions are bit-identical to the stand-alone launch's (tests: native step driver ==
// method-by-method path).  The group's actions go to act_out (global) and to sAct [16][2] for the env lanes of wave 0.
template <bool PK>
__device__ __forceinline__ void group(const Args& a, int rows, const float* __restrict__ obs, long g, float* smem, float* sAct,
                                      float* __restrict__ act_out) {
    constexpr int IN = 6, OU = 2, XSW = 8;
    float* sA = smem;
    float* sX = sA + A_IMG;
    float* sPart = sX + GROUP * XSW;
    const Lane L;
    const Net net = make_net(a.params, IN, 2 * OU);
    float w2[128];
    SmallRegs<IN, OU> r;
    float xv = 0.f;
    if (threadIdx.x < GROUP * XSW) {
        const int row = threadIdx.x / XSW, i = threadIdx.x % XSW;
        const long gr = g * GROUP + row;
        if (gr < rows && i < IN) xv = obs[gr * IN + i] * a.scale[i];
    }
    float b3v = 0.f;
    if (threadIdx.x < GROUP * OU) b3v = net.b3[threadIdx.x % OU];
    float zmax = 0.f;
    bool saw_nan = false;
    load_small<IN, OU>(net, L, r);
    if constexpr (PK) load_w2_packed(a.pack, L, w2); else load_w2_fwd(net.W2, L, w2);
    saw_nan |= xv != xv;
    if (threadIdx.x < GROUP * XSW) sX[threadIdx.x] = xv;
    lds_barrier();
    float pz = 0.f;
    if (threadIdx.x < GROUP * OU) pz = row_poison(sX + (threadIdx.x / OU) * XSW, XSW);
    float h1[2][4], h2[2][4];
    forward_group<IN, OU>(sX, sA, sPart, L, w2, r, h1, h2, nullptr, 0, nullptr, &zmax);
    const int tid = threadIdx.x;
    if (tid < GROUP * OU) {
        const int row = (tid / OU) % GROUP, o = tid % OU;
        const long gr = g * GROUP + row;
        float y = 0.f;
        if (gr < rows) {
            float z = out_preact(sPart, b3v, row, o);
            y = a.out_tanh ? a.out_scale * tanhf(z) : z;
            y += pz;
            if (a.sigma > 0.f) {   // OffPolicyWorker.sample: action += N(0, sigma), worker.py:97-98
                Philox4 p = philox4x32_10((uint32_t)gr, a.c1, a.c2, 0x5eedu + (uint32_t)o, a.k0, a.k1);
                float u1 = u01(p.v[0]), u2 = u01(p.v[1]);
                y = add_gauss_noise(y, a.sigma, u1, u2);
            }
            saw_nan |= y != y;
            act_out[gr * OU + o] = y;
        }
        sAct[tid] = y;
    }
    report_activation_range(a.status, zmax);
    if (a.status && saw_nan) atomicOr(a.status, MPG_STATUS_NAN);
}
}  // namespace worker_policy
#pragma clang fp contract(off)

namespace {

#ifdef MPG_TIMELINE   // diagnostic build only (tools/timeline.sh)
__device__ unsigned long long g_env_tl[2][16];
#define ENV_TL(k) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 200)) g_env_tl[blockIdx.x ? 1 : 0][k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ENV_TL(k)
#endif

constexpr float PI_F = 3.14159265358979323846f;          // float32(np.pi)
constexpr float TWO_PI_F = (float)(2.0 * 3.14159265358979323846);   // float32(2*np.pi)
constexpr float PERIOD = 1200.f;                          // path_tracking_env.py:205

// vehicle parameters, path_tracking_env.py:60-68 (cast to float32 like :86-93)
constexpr float C_f = -128915.5f, C_r = -85943.6f, A_ = 1.06f, B_ = 1.85f, MASS = 1412.f, I_z = 1536.7f,
                MIU = 1.0f, G_ = 9.81f;

struct PathRef {
    float y, phi;
};

// sin and cos of a bounded argument (headings within +-pi, path phases below 38 rad) without the library routine's
// large-argument machinery: three-term Cody-Waite reduction by pi/2 (k * 1.5703125 is exact for |k| < 2^8), then the
// minimax polynomials of the Cephes single-precision kernels on [-pi/4, pi/4].  Measured against float64 on 4M draws
// each of [-3.3, 3.3], [0, 38] and [-200, 200]: absolute error <= 9.2e-8 for both (numpy's float32 sin/cos: 8.5e-8);
// the results only ever enter sums with O(1) terms.  ~25 instructions for both, against ~110 for sincosf - the env
// kernel spends most of its time in the 20 + 3 of them per step.  |a| > 200 (a caller's own far-away x) takes sincosf.
__device__ __forceinline__ void sincos_bounded(float a, float& s, float& c) {
    if (fabsf(a) > 200.f) {
        sincosf(a, &s, &c);
        return;
    }
    const float kf = rintf(a * 0.63661977236758134308f);            // a * 2/pi
    float r = fmaf(-kf, 1.5703125f, a);
    r = fmaf(-kf, 4.837512969970703125e-4f, r);
    r = fmaf(-kf, 7.54978995489188216e-8f, r);
    const float z = r * r;
    const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                          fmaf(-0.5f, z, 1.f));
    const int q = (int)kf;
    const float ss = (q & 1) ? pc : ps, cc = (q & 1) ? ps : pc;
    s = (q & 2) ? -ss : ss;
    c = ((q + 1) & 2) ? -cc : cc;
}

// ReferencePath.compute_path_y / compute_path_phi, path_tracking_env.py:207-220.
// numpy evaluates (x - shift) * 2 * np.pi / T in float32, one rounding per operator; y and the slope are
// accumulated curve by curve into a float32 zero array.
__device__ __forceinline__ PathRef path_ref(float x) {
    const float Amp[3] = {7.5f, 2.5f, -5.f};
    const float T[3] = {200.f, 300.f, 400.f};
    float y = 0.f, d = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float arg = (((x - 0.f) * 2.f) * PI_F) / T[i];
        float s, c;
        sincos_bounded(arg, s, c);
        y = y + Amp[i] * s;
        // magnitude * 2 * np.pi / T is a python-float (double) scalar, rounded once when it meets the array
        const float k = (float)((double)Amp[i] * 2.0 * 3.14159265358979323846 / (double)T[i]);
        d = d + k * c;
    }
    PathRef r;
    r.y = y;
    r.phi = atanf(d);
    return r;
}

// x / 200.f, correctly rounded, in three instructions instead of the ~12 of an IEEE division: q = RN(x * RN(1/200)), the
// exact remainder by one fma, one correction fma (Markstein).  Checked against the division on every float32 of 30 binades
// (2^-17 .. 2^12, 2.5e8 values, no mismatch; outside the denormal range the mantissa behaviour does not depend on the
// binade).  Three of the five divisions of a sub-step are by 200.
__device__ __forceinline__ float div200(float x) {
    const float c = 0.005f;                    // RN(1/200)
    const float q = x * c;
    const float r = fmaf(-q, 200.f, x);
    return fmaf(r, c, q);
}

__device__ __forceinline__ float wrap_pi(float a) {        // :168-169 / :176-177
    if (a > PI_F) a = a - TWO_PI_F;
    if (a <= -PI_F) a = a + TWO_PI_F;
    return a;
}

__global__ void __launch_bounds__(64) k_reset_from_obs(int n, int od, float* __restrict__ st, const float* __restrict__ obs) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* o = obs + (size_t)i * od;                                            // only the six base entries are read
    float vx = o[0] + 20.f, vy = o[1], r = o[2], dy = o[3], dphi = o[4], x = o[5];   // _get_state :404-408
    PathRef p = path_ref(x);                                                          // :415-417
    st[0 * (size_t)n + i] = vx;
    st[1 * (size_t)n + i] = vy;
    st[2 * (size_t)n + i] = r;
    st[3 * (size_t)n + i] = dy + p.y;      // :420
    st[4 * (size_t)n + i] = dphi + p.phi;  // :419 (no wrap here, as in the reference)
    st[5 * (size_t)n + i] = x;
    st[6 * (size_t)n + i] = dy;
    st[7 * (size_t)n + i] = dphi;
}

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
    float u1 = u01(a), u2 = u01(b);
    float rad = sqrtf(-2.f * logf(u1));
    float s, c;
    sincos_bounded(TWO_PI_F * u2, s, c);                  // argument in [0, 2 pi)
    z0 = rad * c;
    z1 = rad * s;
}

struct Agent {   // one agent's state block in registers
    float vx, vy, r, y, phi, x, dy, dphi;
};

__device__ __forceinline__ Agent load_agent(const float* __restrict__ st, size_t N, int i) {
    Agent a;
    a.vx = st[0 * N + i]; a.vy = st[1 * N + i]; a.r = st[2 * N + i]; a.y = st[3 * N + i];
    a.phi = st[4 * N + i]; a.x = st[5 * N + i]; a.dy = st[6 * N + i]; a.dphi = st[7 * N + i];
    return a;
}
__device__ __forceinline__ void store_agent(float* __restrict__ st, size_t N, int i, const Agent& a) {
    st[0 * N + i] = a.vx; st[1 * N + i] = a.vy; st[2 * N + i] = a.r; st[3 * N + i] = a.y;
    st[4 * N + i] = a.phi; st[5 * N + i] = a.x; st[6 * N + i] = a.dy; st[7 * N + i] = a.dphi;
}
// ReferencePath.compute_path_y alone (the look-ahead terms need no heading)
__device__ __forceinline__ float path_y_only(float x) {
    const float Amp[3] = {7.5f, 2.5f, -5.f};
    const float T[3] = {200.f, 300.f, 400.f};
    float y = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) y = y + Amp[i] * sinf((((x - 0.f) * 2.f) * PI_F) / T[i]);
    return y;
}

// _get_obs, path_tracking_env.py:385-402: [v_x - 20, v_y, r, delta_y, delta_phi, x | K look-ahead delta-y terms].
// od = 6 + K.  The look-ahead abscissa advances by v_x * 1. / 200 * 20 * 2 per entry (float32, operator by operator,
// :394-395) and is NOT wrapped to the path period; y is the vehicle's current world-frame y.
__device__ __forceinline__ void write_obs(float* __restrict__ obs, int i, int od, const Agent& a) {
    if (od == 6) {
        float2* o = reinterpret_cast<float2*>(obs + (size_t)i * 6);
        o[0] = make_float2(a.vx - 20.f, a.vy);
        o[1] = make_float2(a.r, a.dy);
        o[2] = make_float2(a.dphi, a.x);
        return;
    }
    float* o = obs + (size_t)i * od;
    o[0] = a.vx - 20.f; o[1] = a.vy; o[2] = a.r; o[3] = a.dy; o[4] = a.dphi; o[5] = a.x;
    float x_ = a.x;
    const float adv = (((a.vx * 1.f) / 200.f) * 20.f) * 2.f;
    for (int k = 6; k < od; ++k) {
        x_ = x_ + adv;
        o[k] = a.y - path_y_only(x_);
    }
}

// PathTrackingEnv.reset() for one agent, path_tracking_env.py:423-454, on the Philox stream (i, ctr)
__device__ __forceinline__ void reset_agent(Agent& ag, int i, uint32_t k0, uint32_t k1, uint32_t c1, uint32_t c2) {
    Philox4 a = philox4x32_10((uint32_t)i, c1, c2, 0u, k0, k1);
    Philox4 b = philox4x32_10((uint32_t)i, c1, c2, 1u, k0, k1);
    float x = 600.f * u01(a.v[0]);                    // :426
    float vx = 15.f + 10.f * u01(a.v[1]);             // :434
    float z0, z1, z2, z3;
    box_muller(a.v[2], a.v[3], z0, z1);
    box_muller(b.v[0], b.v[1], z2, z3);
    float dy = z0;                                     // :428
    float dphi = z1 * (float)(3.14159265358979323846 / 9.0);   // :431
    float beta = z2 * 0.15f;                           // :435
    float r = z3 * 0.3f;                               // :437
    PathRef p = path_ref(x);
    float y = dy + p.y;                                // compute_y :222-224
    float phi = wrap_pi(dphi + p.phi);                 // compute_phi :230-235
    float sb, cb;                                       // beta = 0.15 * N(0, 1): far inside the bounded range
    sincos_bounded(beta, sb, cb);
    ag.vx = vx; ag.vy = vx * (sb / cb);                // :436, tan(beta)
    ag.r = r; ag.y = y; ag.phi = phi; ag.x = x;
    ag.dy = y - p.y;                                   // :450
    ag.dphi = phi - p.phi;                             // :449
}

struct StepOut {
    float reward;
    bool done, done_intended;
};

// PathTrackingEnv.step for one agent (:456-487): reward on the pre-step state, 20 sub-steps, done flags
__device__ __forceinline__ StepOut step_agent(Agent& ag, const float2 an) {
    StepOut out;
    float vx = ag.vx, vy = ag.vy, r = ag.r, y = ag.y, phi = ag.phi, x = ag.x, dy = ag.dy, dphi = ag.dphi;
    // step(): scale and clip the action, path_tracking_env.py:457-459
    const float ACT_HI0 = (float)(1.2 * 3.14159265358979323846 / 9.0), ACT_HI1 = 3.f;
    float steer = ((an.x * 1.2f) * PI_F) / 9.f;
    float a_x = an.y * 3.f;
    steer = fminf(fmaxf(steer, -ACT_HI0), ACT_HI0);
    a_x = fminf(fmaxf(a_x, -ACT_HI1), ACT_HI1);

    // compute_rewards on the PRE-step veh_state, :181-199
    {
        float t = vx - 20.f;
        float devi_v = -(t * t), devi_y = -(dy * dy), devi_phi = -(dphi * dphi), p_yaw = -(r * r),
              p_steer = -(steer * steer), p_ax = -(a_x * a_x);
        out.reward = 0.01f * devi_v + 0.04f * devi_y + 0.1f * devi_phi + 0.02f * p_yaw + 5.f * p_steer + 0.05f * p_ax;
    }

    // f_xu pieces that depend on the action only, :95-101,135-136
    const float F_zf = B_ * MASS * G_ / (A_ + B_), F_zr = A_ * MASS * G_ / (A_ + B_);
    const float F_xf = a_x < 0.f ? MASS * a_x / 2.f : 0.f;
    const float F_xr = a_x < 0.f ? MASS * a_x / 2.f : MASS * a_x;
    const float miu_f = sqrtf((MIU * F_zf) * (MIU * F_zf) - F_xf * F_xf) / F_zf;
    const float miu_r = sqrtf((MIU * F_zr) * (MIU * F_zr) - F_xr * F_xr) / F_zr;

    const float tau = 0.005f;                  // 1/base_freq as a python float, cast on contact (:141)
    const float K1 = tau * (A_ * C_f - B_ * C_r);
    const float K2 = tau * C_f, K3 = tau * MASS, K4 = tau * (C_f + C_r);
    const float K5 = (tau * A_) * C_f;
    const float K6 = tau * ((A_ * A_) * C_f + (B_ * B_) * C_r);

    // simulation(), :144-179.  delta_y / delta_phi are overwritten by every sub-step, so the reference path is evaluated once,
    // for the last one.  The loop is deliberately NOT unrolled: one wave per CU runs this kernel, once per launch, with a cold
    // instruction cache - 20 unrolled sub-steps (7300 instructions, 58 KB) spent more time fetching code than executing
    // it (18.5 us -> 22 us when the unrolled body was made cheaper; measured).
    float vx_pre = vx, vy_pre = vy, r_pre = r;  // state entering the LAST sub-step (for `others`)
    float x_u = x, phi_u = phi;
#pragma unroll 1
    for (int s = 0; s < 20; ++s) {
        vx_pre = vx; vy_pre = vy; r_pre = r;
        // prediction -> f_xu with tau = 1/200 on (v_x, v_y, r); entries 3-5 are overwritten below
        float nvx = vx + tau * (a_x + vy * r);
        float nvy = (((MASS * vy) * vx + K1 * r) - (K2 * steer) * vx - (K3 * (vx * vx)) * r) / (MASS * vx - K4);
        float nr = ((((-I_z) * r) * vx - K1 * vy) + (K5 * steer) * vx) / (K6 - I_z * vx);
        nvx = fminf(fmaxf(nvx, 1.f), 35.f);     // :153
        // world frame, :156-160: phi first, then y and x with the OLD v_x, v_y but the NEW phi (view aliasing)
        phi = phi + div200(r);
        float sp, cp;
        sincos_bounded(phi, sp, cp);
        y = y + div200(vx * sp + vy * cp);
        x = x + div200(vx * cp - vy * sp);
        vx = nvx; vy = nvy; r = nr;             // :161
        x_u = x; phi_u = phi;                   // :163-165 read x and phi before their wraps
        phi = wrap_pi(phi);                     // :168-169
        if (x > PERIOD) x = x - PERIOD;         // :171
        if (x <= 0.f) x = x + PERIOD;           // :172
    }
    ENV_TL(5);
    {
        PathRef p = path_ref(x_u);              // :163-164
        dphi = wrap_pi(phi_u - p.phi);          // :165, :176-177
        dy = y - p.y;                           // :166
    }
    ENV_TL(6);

    // `others` of the last sub-step (:100-101,135-138), judge_done :474-487
    const float alpha_f = atanf((vy_pre + A_ * r_pre) / vx_pre) - steer;
    const float alpha_r = atanf((vy_pre - B_ * r_pre) / vx_pre);
    const float afb = 3.f * miu_f * F_zf / C_f, arb = 3.f * miu_r * F_zr / C_r;
    const float rb = miu_r * G_ / fabsf(vx_pre);
    const bool geo = (fabsf(dy) > 3.f) | (fabsf(dphi) > PI_F / 4.f) | (vx < 2.f);
    const bool lit = geo | (alpha_f < -afb) | (alpha_f > afb) | (alpha_r < -arb) | (alpha_r > arb) | (r < -rb) |
                     (r > rb);
    out.done = lit;
    out.done_intended = geo | (fabsf(alpha_f) > fabsf(afb)) | (fabsf(alpha_r) > fabsf(arb)) | (fabsf(r) > rb);

    ag.vx = vx; ag.vy = vy; ag.r = r; ag.y = y; ag.phi = phi; ag.x = x; ag.dy = dy; ag.dphi = dphi;
    return out;
}

__global__ void __launch_bounds__(64) k_reset(int n, float* __restrict__ st, const uint8_t* __restrict__ mask,
                                              uint32_t k0, uint32_t k1, uint32_t c1, uint32_t c2,
                                              float* __restrict__ obs, int od) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Agent ag = load_agent(st, n, i);
    if (mask == nullptr || mask[i]) {
        reset_agent(ag, i, k0, k1, c1, c2);
        store_agent(st, n, i, ag);
    }
    write_obs(obs, i, od, ag);
}

__global__ void __launch_bounds__(64) k_step(int n, float* __restrict__ st, const float* __restrict__ action,
                                             float* __restrict__ obs, float* __restrict__ reward,
                                             uint8_t* __restrict__ done, uint8_t* __restrict__ done_intended, int od) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Agent ag = load_agent(st, n, i);
    const StepOut o = step_agent(ag, reinterpret_cast<const float2*>(action)[i]);
    reward[i] = o.reward;
    done[i] = o.done ? 1 : 0;
    if (done_intended) done_intended[i] = o.done_intended ? 1 : 0;
    store_agent(st, n, i, ag);
    write_obs(obs, i, od, ag);
}

// OffPolicyWorker.sample's inner body after the policy (worker.py:108-112) in one launch: env.step, the transition
// (obs, action, RAW reward, obs', done) written straight into the replay ring slot (next_idx + i) % capacity
// (buffer.py:46-55), then env.reset() for the agents that are done (path_tracking_env.py:445).
// The same step with FOUR lanes per agent (lanes 4a .. 4a+3 of one wave, q = lane & 3; k_step_store_reset).  The sub-steps'
// serial part is the planar dynamics (v_x, v_y, r) and the heading; the sincos of each heading and the world-frame
// increments are not on that chain and delta_y / delta_phi only matter after the last sub-step.  So: (A) every lane runs the
// short serial chain and the 20 (heading, v_x, v_y) triples go to LDS, (B) lane q evaluates the increments of sub-steps
// q, q+4, .. (5 of the 20 sincos each), (C) every lane adds the increments in order.  Same float32 operations on the same
// operands in the same order per variable as step_agent: bit-identical results, ~2450 instead of ~4300 instructions per wave,
// and 256 waves instead of 64 for 4096 agents.  sc: this agent's 100 floats of LDS, [5][20].
__device__ __forceinline__ StepOut step_agent_quad(Agent& ag, const float2 an, const int q, float* sc) {
    StepOut out;
    float vx = ag.vx, vy = ag.vy, r = ag.r, y = ag.y, phi = ag.phi, x = ag.x, dy = ag.dy, dphi = ag.dphi;
    // step(): scale and clip the action, path_tracking_env.py:457-459
    const float ACT_HI0 = (float)(1.2 * 3.14159265358979323846 / 9.0), ACT_HI1 = 3.f;
    float steer = ((an.x * 1.2f) * PI_F) / 9.f;
    float a_x = an.y * 3.f;
    steer = fminf(fmaxf(steer, -ACT_HI0), ACT_HI0);
    a_x = fminf(fmaxf(a_x, -ACT_HI1), ACT_HI1);

    // compute_rewards on the PRE-step veh_state, :181-199
    {
        float t = vx - 20.f;
        float devi_v = -(t * t), devi_y = -(dy * dy), devi_phi = -(dphi * dphi), p_yaw = -(r * r),
              p_steer = -(steer * steer), p_ax = -(a_x * a_x);
        out.reward = 0.01f * devi_v + 0.04f * devi_y + 0.1f * devi_phi + 0.02f * p_yaw + 5.f * p_steer + 0.05f * p_ax;
    }

    // f_xu pieces that depend on the action only, :95-101,135-136
    const float F_zf = B_ * MASS * G_ / (A_ + B_), F_zr = A_ * MASS * G_ / (A_ + B_);
    const float F_xf = a_x < 0.f ? MASS * a_x / 2.f : 0.f;
    const float F_xr = a_x < 0.f ? MASS * a_x / 2.f : MASS * a_x;
    const float miu_f = sqrtf((MIU * F_zf) * (MIU * F_zf) - F_xf * F_xf) / F_zf;
    const float miu_r = sqrtf((MIU * F_zr) * (MIU * F_zr) - F_xr * F_xr) / F_zr;

    const float tau = 0.005f;                  // 1/base_freq as a python float, cast on contact (:141)
    const float K1 = tau * (A_ * C_f - B_ * C_r);
    const float K2 = tau * C_f, K3 = tau * MASS, K4 = tau * (C_f + C_r);
    const float K5 = (tau * A_) * C_f;
    const float K6 = tau * ((A_ * A_) * C_f + (B_ * B_) * C_r);

    float vx_pre = vx, vy_pre = vy, r_pre = r;  // state entering the LAST sub-step (for `others`)
    float phi_u = phi;
    ENV_TL(2);
#pragma unroll 1
    for (int s = 0; s < 20; ++s) {              // (A) simulation(), :144-179: dynamics and heading
        vx_pre = vx; vy_pre = vy; r_pre = r;
        float nvx = vx + tau * (a_x + vy * r);
        float nvy = (((MASS * vy) * vx + K1 * r) - (K2 * steer) * vx - (K3 * (vx * vx)) * r) / (MASS * vx - K4);
        float nr = ((((-I_z) * r) * vx - K1 * vy) + (K5 * steer) * vx) / (K6 - I_z * vx);
        nvx = fminf(fmaxf(nvx, 1.f), 35.f);     // :153
        phi = phi + div200(r);                  // :156, the OLD yaw rate
        if ((s & 3) == q) { sc[s] = phi; sc[20 + s] = vx; sc[40 + s] = vy; }   // new heading, OLD velocities (:157-160)
        phi_u = phi;
        vx = nvx; vy = nvy; r = nr;             // :161
        phi = wrap_pi(phi);                     // :168-169
    }
    __builtin_amdgcn_wave_barrier();            // the four lanes of an agent sit in one wave: LDS is in order within it
    ENV_TL(3);
#pragma unroll 1
    for (int s = q; s < 20; s += 4) {           // (B) this lane's five sub-steps
        float sp, cp;
        sincos_bounded(sc[s], sp, cp);
        const float ovx = sc[20 + s], ovy = sc[40 + s];
        sc[60 + s] = div200(ovx * sp + ovy * cp);
        sc[80 + s] = div200(ovx * cp - ovy * sp);
    }
    __builtin_amdgcn_wave_barrier();
    ENV_TL(4);
    float x_u = x;
#pragma unroll 1
    for (int s4 = 0; s4 < 20; s4 += 4) {        // (C) positions, in order; four increments per LDS round trip
        const float4 iy = *reinterpret_cast<const float4*>(sc + 60 + s4), ix = *reinterpret_cast<const float4*>(sc + 80 + s4);
        const float ay[4] = {iy.x, iy.y, iy.z, iy.w}, ax[4] = {ix.x, ix.y, ix.z, ix.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            y = y + ay[u];
            x = x + ax[u];
            x_u = x;                            // :163-164 read x before the wrap
            if (x > PERIOD) x = x - PERIOD;     // :171
            if (x <= 0.f) x = x + PERIOD;       // :172
        }
    }
    ENV_TL(5);
    {
        PathRef p = path_ref(x_u);              // :163-164
        dphi = wrap_pi(phi_u - p.phi);          // :165, :176-177
        dy = y - p.y;                           // :166
    }
    ENV_TL(6);

    // `others` of the last sub-step (:100-101,135-138), judge_done :474-487
    const float alpha_f = atanf((vy_pre + A_ * r_pre) / vx_pre) - steer;
    const float alpha_r = atanf((vy_pre - B_ * r_pre) / vx_pre);
    const float afb = 3.f * miu_f * F_zf / C_f, arb = 3.f * miu_r * F_zr / C_r;
    const float rb = miu_r * G_ / fabsf(vx_pre);
    const bool geo = (fabsf(dy) > 3.f) | (fabsf(dphi) > PI_F / 4.f) | (vx < 2.f);
    const bool lit = geo | (alpha_f < -afb) | (alpha_f > afb) | (alpha_r < -arb) | (alpha_r > arb) | (r < -rb) |
                     (r > rb);
    out.done = lit;
    out.done_intended = geo | (fabsf(alpha_f) > fabsf(afb)) | (fabsf(alpha_r) > fabsf(arb)) | (fabsf(r) > rb);

    ag.vx = vx; ag.vy = vy; ag.r = r; ag.y = y; ag.phi = phi; ag.x = x; ag.dy = dy; ag.dphi = dphi;
    return out;
}

struct RingPtrs {
    float *obs, *act, *rew, *obs2;
    uint8_t* done;
};
// The next minibatch draw, gathered by spare workgroups of the env launch (mpg_env_step_store_reset_draw): one lane per
// drawn row, the same Philox draw as k_target_fused / k_sample_gather; rows whose slot the env lanes are writing are left
// to the consumer.
struct PreDraw {
    int rows, n_storage, env_blocks;
    uint32_t k0, k1, c1, c2;
    int* o_idx;
    float *o_obs, *o_act, *o_rew, *o_obs2, *o_done;
};
__device__ __forceinline__ void predraw_row(const PreDraw& d, const RingPtrs& ring, int capacity, int fresh_start, int fresh_count,
                                            int gr) {
    const Philox4 p = philox4x32_10((uint32_t)(gr >> 2), d.c1, d.c2, 0x1d5u, d.k0, d.k1);
    const long sr = (long)(((uint64_t)philox_word(p, gr & 3) * (uint64_t)d.n_storage) >> 32);
    int off = (int)sr - fresh_start;
    if (off < 0) off += capacity;
    if (off < fresh_count) return;
    float o1[6], o2[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { o1[i] = ring.obs[sr * 6 + i]; o2[i] = ring.obs2[sr * 6 + i]; }
    const float2 ac = reinterpret_cast<const float2*>(ring.act)[sr];
    const float rw = ring.rew[sr];
    const uint8_t dn = ring.done[sr];
    if (d.o_idx) d.o_idx[gr] = (int)sr;
#pragma unroll
    for (int i = 0; i < 6; ++i) { d.o_obs[(long)gr * 6 + i] = o1[i]; d.o_obs2[(long)gr * 6 + i] = o2[i]; }
    reinterpret_cast<float2*>(d.o_act)[gr] = ac;
    d.o_rew[gr] = rw;
    if (d.o_done) d.o_done[gr] = (float)dn;
}

__global__ void __launch_bounds__(64) k_step_store_reset(int n, float* __restrict__ st, const float* __restrict__ action,
                                                         RingPtrs ring, int capacity, int next_idx, uint32_t k0, uint32_t k1,
                                                         uint32_t c1, uint32_t c2, float* __restrict__ obs_out,
                                                         uint8_t* __restrict__ done_out, int od, PreDraw pd) {
    if (pd.rows > 0 && (int)blockIdx.x >= pd.env_blocks) {
        const int gr = ((int)blockIdx.x - pd.env_blocks) * 64 + threadIdx.x;
        if (gr < pd.rows) predraw_row(pd, ring, capacity, next_idx, n, gr);
        return;
    }
    __shared__ __attribute__((aligned(16))) float s_quad[16 * 100];
    ENV_TL(0);
    const int t = blockIdx.x * blockDim.x + threadIdx.x, i = t >> 2, q = t & 3;
    if (i >= n) return;
    Agent ag = load_agent(st, n, i);
    const float2 an = reinterpret_cast<const float2*>(action)[i];
    const size_t slot = (size_t)((next_idx + i) % capacity);
    if (q == 0) {
        write_obs(ring.obs, (int)slot, od, ag);                 // obs before the step
        reinterpret_cast<float2*>(ring.act)[slot] = an;
    }
    ENV_TL(1);
    const StepOut o = step_agent_quad(ag, an, q, s_quad + (threadIdx.x >> 2) * 100);
    ENV_TL(7);
    if (q == 1) {
        write_obs(ring.obs2, (int)slot, od, ag);
        ring.rew[slot] = o.reward;
        ring.done[slot] = o.done ? 1 : 0;
        if (done_out) done_out[i] = o.done ? 1 : 0;
    }
    if (o.done) reset_agent(ag, i, k0, k1, c1, c2);
    if (q == 2) store_agent(st, n, i, ag);
    if (q == 3) write_obs(obs_out, i, od, ag);
    ENV_TL(8);
}

// The same launch with ONE lane per agent (step_agent, like k_step): the four-lane form above buys latency at 4096 agents (256 waves
// instead of 64 on 256 CUs) at the price of ~2.3x the instructions per agent; from MPG_ENV_ONE_LANE_FROM agents on the chip is full either
// way and the instruction count is what is left (2^20 agents: 264 us four-lane against k_step's 61).  step_agent and step_agent_quad
// perform the same float32 operations on the same operands in the same order per variable, so the two forms are bit-identical
// (tests/test_env_gpu.py compares fused and separate calls at both sizes).
constexpr int MPG_ENV_ONE_LANE_FROM = 65536;
__global__ void __launch_bounds__(64) k_step_store_reset_1(int n, float* __restrict__ st, const float* __restrict__ action,
                                                           RingPtrs ring, int capacity, int next_idx, uint32_t k0, uint32_t k1,
                                                           uint32_t c1, uint32_t c2, float* __restrict__ obs_out,
                                                           uint8_t* __restrict__ done_out, int od, PreDraw pd) {
    if (pd.rows > 0 && (int)blockIdx.x >= pd.env_blocks) {
        const int gr = ((int)blockIdx.x - pd.env_blocks) * 64 + threadIdx.x;
        if (gr < pd.rows) predraw_row(pd, ring, capacity, next_idx, n, gr);
        return;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Agent ag = load_agent(st, n, i);
    const float2 an = reinterpret_cast<const float2*>(action)[i];
    const size_t slot = (size_t)((next_idx + i) % capacity);
    write_obs(ring.obs, (int)slot, od, ag);                     // obs before the step
    reinterpret_cast<float2*>(ring.act)[slot] = an;
    const StepOut o = step_agent(ag, an);
    write_obs(ring.obs2, (int)slot, od, ag);
    ring.rew[slot] = o.reward;
    ring.done[slot] = o.done ? 1 : 0;
    if (done_out) done_out[i] = o.done ? 1 : 0;
    if (o.done) reset_agent(ag, i, k0, k1, c1, c2);
    store_agent(st, n, i, ag);
    write_obs(obs_out, i, od, ag);
}

// OffPolicyWorker.sample's whole inner body (worker.py:95-112) in ONE launch: the policy pass of a 16-agent group by a 512-thread
// workgroup, then env.step -> ring -> env.reset of those 16 agents by the four-lane form on wave 0 (k_step_store_reset's body).  The
// stand-alone pair is two launches of one wave per CU each (7 + 14 us at 4096 agents); fused, the env lanes start the moment their
// group's actions exist.  Blocks beyond the policy groups gather the minibatch about to be drawn (predraw_row).
template <bool PK>
__global__ void __launch_bounds__(mlp::NTHREAD, 2) k_policy_step_store_reset(const worker_policy::Args pa, int n, float* __restrict__ st,
                                                                              float* __restrict__ obs_io, float* __restrict__ act_out,
                                                                              RingPtrs ring, int capacity, int next_idx, uint32_t k0,
                                                                              uint32_t k1, uint32_t c1, uint32_t c2,
                                                                              uint8_t* __restrict__ done_out, PreDraw pd) {
    if ((int)blockIdx.x >= pd.env_blocks) {
        const int gr = ((int)blockIdx.x - pd.env_blocks) * mlp::NTHREAD + threadIdx.x;
        if (gr < pd.rows) predraw_row(pd, ring, capacity, next_idx, n, gr);
        return;
    }
    __shared__ __attribute__((aligned(16))) float smem[worker_policy::SMEM_FLOATS];
    __shared__ __attribute__((aligned(16))) float s_quad[16 * 100];
    __shared__ float sAct[mlp::GROUP * 2];
    worker_policy::group<PK>(pa, n, obs_io, blockIdx.x, smem, sAct, act_out);
    if (threadIdx.x >= 64) return;                     // the env lanes: wave 0, four lanes per agent (it wrote sAct itself: LDS is in
    __builtin_amdgcn_wave_barrier();                   // order within a wave)
    const int i = blockIdx.x * mlp::GROUP + (threadIdx.x >> 2), q = threadIdx.x & 3;
    if (i >= n) return;
    Agent ag = load_agent(st, n, i);
    const float2 an = make_float2(sAct[2 * (threadIdx.x >> 2)], sAct[2 * (threadIdx.x >> 2) + 1]);
    const size_t slot = (size_t)((next_idx + i) % capacity);
    if (q == 0) {
        write_obs(ring.obs, (int)slot, 6, ag);                  // obs before the step
        reinterpret_cast<float2*>(ring.act)[slot] = an;
    }
    const StepOut o = step_agent_quad(ag, an, q, s_quad + (threadIdx.x >> 2) * 100);
    if (q == 1) {
        write_obs(ring.obs2, (int)slot, 6, ag);
        ring.rew[slot] = o.reward;
        ring.done[slot] = o.done ? 1 : 0;
        if (done_out) done_out[i] = o.done ? 1 : 0;
    }
    if (o.done) reset_agent(ag, i, k0, k1, c1, c2);
    if (q == 2) store_agent(st, n, i, ag);
    if (q == 3) write_obs(obs_io, i, 6, ag);
}

inline bool pt_obs_dim_ok(int od) { return od >= 6 && od <= 6 + MPG_ENV_MAX_FUTURE; }

}  // namespace

extern "C" int mpg_env_reset_from_obs(int env_kind, int n, int obs_dim, float* state, const float* init_obs, mpg_stream_t stream) {
    if (env_kind == MPG_ENV_INVERTED_PENDULUM) return cart_pole::reset_from_obs(n, obs_dim, state, init_obs, mpg_stream(stream));
    MPG_REQUIRE(env_kind == MPG_ENV_PATH_TRACKING, "mpg_env_reset_from_obs: unknown env kind %d", env_kind);
    MPG_REQUIRE(n > 0 && state && init_obs && pt_obs_dim_ok(obs_dim), "mpg_env_reset_from_obs: bad argument");
    hipLaunchKernelGGL(k_reset_from_obs, dim3((n + 63) / 64), dim3(64), 0, mpg_stream(stream), n, obs_dim, state, init_obs);
    MPG_CHECK_LAUNCH("mpg_env_reset_from_obs");
    return MPG_OK;
}

extern "C" int mpg_env_reset(int env_kind, int n, int obs_dim, float* state, const uint8_t* done_mask, uint64_t seed, uint64_t ctr,
                             float* obs, mpg_stream_t stream) {
    if (env_kind == MPG_ENV_INVERTED_PENDULUM) return cart_pole::reset(n, obs_dim, state, done_mask, seed, ctr, obs, mpg_stream(stream));
    MPG_REQUIRE(env_kind == MPG_ENV_PATH_TRACKING, "mpg_env_reset: unknown env kind %d", env_kind);
    MPG_REQUIRE(n > 0 && state && obs && pt_obs_dim_ok(obs_dim), "mpg_env_reset: bad argument");
    hipLaunchKernelGGL(k_reset, dim3((n + 63) / 64), dim3(64), 0, mpg_stream(stream), n, state, done_mask,
                       (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32), obs, obs_dim);
    MPG_CHECK_LAUNCH("mpg_env_reset");
    return MPG_OK;
}

extern "C" int mpg_env_step(int env_kind, int n, int obs_dim, float* state, const float* action, float* obs, float* reward,
                            uint8_t* done, uint8_t* done_intended, mpg_stream_t stream) {
    if (env_kind == MPG_ENV_INVERTED_PENDULUM)
        return cart_pole::step(n, obs_dim, state, action, obs, reward, done, done_intended, mpg_stream(stream));
    MPG_REQUIRE(env_kind == MPG_ENV_PATH_TRACKING, "mpg_env_step: unknown env kind %d", env_kind);
    MPG_REQUIRE(n > 0 && state && action && obs && reward && done && pt_obs_dim_ok(obs_dim), "mpg_env_step: bad argument");
    hipLaunchKernelGGL(k_step, dim3((n + 63) / 64), dim3(64), 0, mpg_stream(stream), n, state, action, obs, reward,
                       done, done_intended, obs_dim);
    MPG_CHECK_LAUNCH("mpg_env_step");
    return MPG_OK;
}

namespace {
int step_store_reset_impl(int env_kind, int n, int obs_dim, float* state, const float* action, int capacity, int next_idx,
                          float* ring_obs, float* ring_act, float* ring_rew, float* ring_obs2, uint8_t* ring_done, uint64_t seed,
                          uint64_t ctr, float* obs_out, uint8_t* done_out, const PreDraw& pd_in, mpg_stream_t stream) {
    MPG_REQUIRE(n > 0 && state && action && capacity >= n && next_idx >= 0 && next_idx < capacity && ring_obs && ring_act &&
                    ring_rew && ring_obs2 && ring_done && obs_out,
                "mpg_env_step_store_reset: bad argument");
    if (env_kind == MPG_ENV_INVERTED_PENDULUM) {
        MPG_REQUIRE(pd_in.rows == 0, "mpg_env_step_store_reset_draw: path-tracking env only");
        return cart_pole::step_store_reset(n, obs_dim, state, action, capacity, next_idx, ring_obs, ring_act, ring_rew, ring_obs2,
                                           ring_done, seed, ctr, obs_out, done_out, mpg_stream(stream));
    }
    MPG_REQUIRE(env_kind == MPG_ENV_PATH_TRACKING, "mpg_env_step_store_reset: unknown env kind %d", env_kind);
    MPG_REQUIRE(pt_obs_dim_ok(obs_dim), "mpg_env_step_store_reset: obs_dim");
    RingPtrs ring{ring_obs, ring_act, ring_rew, ring_obs2, ring_done};
    PreDraw pd = pd_in;
    if (n >= MPG_ENV_ONE_LANE_FROM) {             // one lane per agent: the throughput form
        pd.env_blocks = (n + 63) / 64;
        const int blocks1 = pd.env_blocks + (pd.rows + 63) / 64;
        hipLaunchKernelGGL(k_step_store_reset_1, dim3(blocks1), dim3(64), 0, mpg_stream(stream), n, state, action, ring,
                           capacity, next_idx, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32), obs_out,
                           done_out, obs_dim, pd);
        MPG_CHECK_LAUNCH("mpg_env_step_store_reset");
        return MPG_OK;
    }
    pd.env_blocks = (4 * n + 63) / 64;            // four lanes per agent: the latency form
    const int blocks = pd.env_blocks + (pd.rows + 63) / 64;
    hipLaunchKernelGGL(k_step_store_reset, dim3(blocks), dim3(64), 0, mpg_stream(stream), n, state, action, ring,
                       capacity, next_idx, (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32), obs_out,
                       done_out, obs_dim, pd);
#ifdef MPG_TIMELINE
    static int s_calls = 0;
    if (++s_calls % 100 == 0) {
        unsigned long long h[2][16];
        (void)hipStreamSynchronize(mpg_stream(stream));
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_env_tl), sizeof(h));
        for (int b = 0; b < 2; ++b) {
            fprintf(stderr, "timeline env wg%d:", b ? 200 : 0);
            for (int k = 1; k < 9; ++k) fprintf(stderr, " %d:%lld", k, (long long)(h[b][k] - h[b][0]));
            fprintf(stderr, "\n");
        }
    }
#endif
    MPG_CHECK_LAUNCH("mpg_env_step_store_reset");
    return MPG_OK;
}
}  // namespace

extern "C" int mpg_env_step_store_reset(int env_kind, int n, int obs_dim, float* state, const float* action, int capacity, int next_idx,
                                        float* ring_obs, float* ring_act, float* ring_rew, float* ring_obs2, uint8_t* ring_done,
                                        uint64_t seed, uint64_t ctr, float* obs_out, uint8_t* done_out, mpg_stream_t stream) {
    PreDraw pd{};
    return step_store_reset_impl(env_kind, n, obs_dim, state, action, capacity, next_idx, ring_obs, ring_act, ring_rew, ring_obs2,
                                 ring_done, seed, ctr, obs_out, done_out, pd, stream);
}

extern "C" int mpg_env_step_store_reset_draw(int env_kind, int n, int obs_dim, float* state, const float* action, int capacity,
                                             int next_idx, float* ring_obs, float* ring_act, float* ring_rew, float* ring_obs2,
                                             uint8_t* ring_done, uint64_t seed, uint64_t ctr, float* obs_out, uint8_t* done_out,
                                             const mpg_replay_draw_t* draw, int rows, float* b_obs, float* b_act, float* b_rew,
                                             float* b_obs2, mpg_stream_t stream) {
    MPG_REQUIRE(env_kind == MPG_ENV_PATH_TRACKING && obs_dim == 6, "mpg_env_step_store_reset_draw: path-tracking env with obs_dim 6 only");
    MPG_REQUIRE(draw && rows > 0 && b_obs && b_act && b_rew && b_obs2 && draw->n_storage > 0 && draw->n_storage <= capacity,
                "mpg_env_step_store_reset_draw: incomplete draw");
    PreDraw pd{};
    pd.rows = rows; pd.n_storage = draw->n_storage;
    pd.k0 = (uint32_t)draw->seed; pd.k1 = (uint32_t)(draw->seed >> 32);
    pd.c1 = (uint32_t)draw->ctr; pd.c2 = (uint32_t)(draw->ctr >> 32);
    pd.o_idx = draw->idx_out; pd.o_done = draw->done_out;
    pd.o_obs = b_obs; pd.o_act = b_act; pd.o_rew = b_rew; pd.o_obs2 = b_obs2;
    return step_store_reset_impl(env_kind, n, obs_dim, state, action, capacity, next_idx, ring_obs, ring_act, ring_rew, ring_obs2,
                                 ring_done, seed, ctr, obs_out, done_out, pd, stream);
}

// worker.py:95-112 for the path-tracking env with six-entry observations: mpg_policy_action + mpg_env_step_store_reset(_draw) as one
// launch (bit-identical actions, ring rows, states and observations).  obs_io [n][6]: the current observations in, the next ones out.
extern "C" int mpg_worker_step(const mpg_cfg_t* cfg, const float* policy_params, int n, float* state, float* obs_io, float explore_sigma,
                               uint64_t noise_seed, uint64_t noise_ctr, float* act_out, int capacity, int next_idx, float* ring_obs,
                               float* ring_act, float* ring_rew, float* ring_obs2, uint8_t* ring_done, uint64_t env_seed,
                               uint64_t env_ctr, uint8_t* done_out, const mpg_replay_draw_t* draw, int rows, float* b_obs, float* b_act,
                               float* b_rew, float* b_obs2, mpg_stream_t stream) {
    MPG_REQUIRE(cfg && cfg->env_kind == MPG_ENV_PATH_TRACKING && cfg->obs_dim == 6 && cfg->act_dim == 2,
                "mpg_worker_step: path-tracking env with obs_dim 6 only");
    MPG_REQUIRE(policy_params && n > 0 && state && obs_io && act_out && capacity >= n && next_idx >= 0 && next_idx < capacity && ring_obs &&
                    ring_act && ring_rew && ring_obs2 && ring_done,
                "mpg_worker_step: bad argument");
    // (the same refusal as mpg_policy_action's cfg_ok: tanh output WITH an action range is not what the reference computes)
    MPG_REQUIRE(!(cfg->policy_out_act == MPG_ACT_TANH && cfg->action_range > 0.f), "mpg_worker_step: tanh policy with an action range");
    PreDraw pd{};
    if (draw) {
        MPG_REQUIRE(rows > 0 && b_obs && b_act && b_rew && b_obs2 && draw->n_storage > 0 && draw->n_storage <= capacity,
                    "mpg_worker_step: incomplete draw");
        pd.rows = rows; pd.n_storage = draw->n_storage;
        pd.k0 = (uint32_t)draw->seed; pd.k1 = (uint32_t)(draw->seed >> 32);
        pd.c1 = (uint32_t)draw->ctr; pd.c2 = (uint32_t)(draw->ctr >> 32);
        pd.o_idx = draw->idx_out; pd.o_done = draw->done_out;
        pd.o_obs = b_obs; pd.o_act = b_act; pd.o_rew = b_rew; pd.o_obs2 = b_obs2;
    }
    pd.env_blocks = (n + mlp::GROUP - 1) / mlp::GROUP;
    worker_policy::Args pa;
    pa.params = policy_params;
    pa.pack = mlp::weight_cache_lookup(cfg, mlp::make_net(policy_params, 6, 4).W2, 0);
    pa.status = mpg_status_of(cfg);
    const bool ranged = cfg->action_range > 0.f;
    pa.out_tanh = (cfg->policy_out_act == MPG_ACT_TANH || ranged) ? 1 : 0;
    pa.out_scale = ranged ? cfg->action_range : 1.f;
    pa.sigma = explore_sigma;
    pa.k0 = (uint32_t)noise_seed; pa.k1 = (uint32_t)(noise_seed >> 32); pa.c1 = (uint32_t)noise_ctr; pa.c2 = (uint32_t)(noise_ctr >> 32);
    for (int i = 0; i < 8; ++i) pa.scale[i] = i < 6 ? cfg->obs_scale[i] : 1.f;
    RingPtrs ring{ring_obs, ring_act, ring_rew, ring_obs2, ring_done};
    const int blocks = pd.env_blocks + (pd.rows + mlp::NTHREAD - 1) / mlp::NTHREAD;
    hipStream_t s = mpg_stream(stream);
    mpg_prof_begin(mpg_prof_of(cfg), 2, s);
    if (pa.pack)
        hipLaunchKernelGGL((k_policy_step_store_reset<true>), dim3(blocks), dim3(mlp::NTHREAD), 0, s, pa, n, state, obs_io, act_out, ring, capacity,
                           next_idx, (uint32_t)env_seed, (uint32_t)(env_seed >> 32), (uint32_t)env_ctr, (uint32_t)(env_ctr >> 32), done_out, pd);
    else
        hipLaunchKernelGGL((k_policy_step_store_reset<false>), dim3(blocks), dim3(mlp::NTHREAD), 0, s, pa, n, state, obs_io, act_out, ring, capacity,
                           next_idx, (uint32_t)env_seed, (uint32_t)(env_seed >> 32), (uint32_t)env_ctr, (uint32_t)(env_ctr >> 32), done_out, pd);
    mpg_prof_end(mpg_prof_of(cfg), 2, s);
    MPG_CHECK_LAUNCH("mpg_worker_step");
    return MPG_OK;
}
