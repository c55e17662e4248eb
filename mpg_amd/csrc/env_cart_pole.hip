// InvertedPendulumConti-v0 as an analytic kernel: the reference steps this environment with MuJoCo
// (envs_and_models/inverted_pendulum_conti.py:5-30 on inverted_pendulum_conti.xml); here the same model - cart on a
// slider, one pole on a hinge, capsule geometry at the default density 1000 kg/m^3, joint damping 1, motor gear 100 with
// ctrlrange +-3, gravity 9.81, integrator RK4, timestep 0.02, frame_skip 2 - is written as the closed-form cart-pole
// equations and integrated with the classical Runge-Kutta scheme, one lane per agent, state in registers.
//
//   (M + m) pdd + m l cos(th) thdd = F - b_x pd + m l sin(th) thd^2
//   m l cos(th) pdd + (I + m l^2) thdd = m g l sin(th) - b_th thd            th measured from the pole's rest axis + th0
//
// PARITY UNPINNED: MuJoCo is not installable here, so no reference run can pin these numbers; the kernel is checked
// against oracle/mpg_oracle.py:InvertedPendulumContiOracle (the float64 statement of the same formulas), and DESIGN.md
// says so.  obs = [p, theta, pdot, thetadot] (:27-28); reward (:12-16); done = not(|p| < 2 and |theta| <= 0.2) (:17-18);
// reset: U(-0.01, 0.01) on all four (:21-25) from Philox(seed, ctr, agent).
//
// State block: rows 0..3 of the opaque [MPG_ENV_STATE_DIM][n] array (rows 4..7 unused).
#include "env_internal.h"

namespace {

constexpr double PI_D = 3.14159265358979323846;
constexpr double RHO = 1000.0;
constexpr double capsule_mass(double r, double h) { return RHO * PI_D * r * r * h + RHO * 4.0 / 3.0 * PI_D * r * r * r; }
constexpr double capsule_inertia(double r, double h) {
    return RHO * PI_D * r * r * h * (h * h / 12 + r * r / 4) +
           RHO * 4.0 / 3.0 * PI_D * r * r * r * (2 * r * r / 5 + h * h / 4 + 3 * h * r / 8);
}
// pole: capsule of radius 0.049 from the hinge to (0.001, 0, 0.6)
constexpr double POLE_LEN = 0.60000083333275462;           // hypot(0.001, 0.6)
constexpr float M_CART = (float)capsule_mass(0.1, 0.2);
constexpr float M_POLE = (float)capsule_mass(0.049, POLE_LEN);
constexpr float L_COM = (float)(0.5 * POLE_LEN);
constexpr float I_POLE = (float)capsule_inertia(0.049, POLE_LEN);
constexpr float TH0 = 0.0016666651234593622f;                 // atan2(0.001, 0.6)
constexpr float G = 9.81f, DT = 0.02f, GEAR = 100.f, CTRL = 3.f, DAMP_X = 1.f, DAMP_TH = 1.f;
constexpr int FRAME_SKIP = 2;

struct S4 {
    float p, th, pd, thd;
};

__device__ __forceinline__ S4 deriv(const S4& s, float force) {
    const float th = s.th + TH0;
    float sn, cs;
    sincosf(th, &sn, &cs);
    const float a11 = M_CART + M_POLE, a12 = M_POLE * L_COM * cs, a22 = I_POLE + M_POLE * L_COM * L_COM;
    const float b1 = force - DAMP_X * s.pd + M_POLE * L_COM * sn * s.thd * s.thd;
    const float b2 = M_POLE * G * L_COM * sn - DAMP_TH * s.thd;
    const float idet = 1.f / (a11 * a22 - a12 * a12);
    S4 d;
    d.p = s.pd;
    d.th = s.thd;
    d.pd = (a22 * b1 - a12 * b2) * idet;
    d.thd = (a11 * b2 - a12 * b1) * idet;
    return d;
}
__device__ __forceinline__ S4 axpy(const S4& s, float h, const S4& k) {
    return S4{s.p + h * k.p, s.th + h * k.th, s.pd + h * k.pd, s.thd + h * k.thd};
}

struct StepOut {
    float reward;
    bool done;
};

__device__ __forceinline__ StepOut step_agent(S4& s, float action) {
    const float force = GEAR * fminf(fmaxf(action, -CTRL), CTRL);
#pragma unroll
    for (int f = 0; f < FRAME_SKIP; ++f) {
        const S4 k1 = deriv(s, force);
        const S4 k2 = deriv(axpy(s, 0.5f * DT, k1), force);
        const S4 k3 = deriv(axpy(s, 0.5f * DT, k2), force);
        const S4 k4 = deriv(axpy(s, DT, k3), force);
        const float h6 = DT / 6.f;
        s.p += h6 * (k1.p + 2.f * k2.p + 2.f * k3.p + k4.p);
        s.th += h6 * (k1.th + 2.f * k2.th + 2.f * k3.th + k4.th);
        s.pd += h6 * (k1.pd + 2.f * k2.pd + 2.f * k3.pd + k4.pd);
        s.thd += h6 * (k1.thd + 2.f * k2.thd + 2.f * k3.thd + k4.thd);
    }
    StepOut o;
    o.reward = -(0.01f * s.p * s.p + s.th * s.th) - (0.1f * s.pd * s.pd + 0.1f * s.thd * s.thd);   // :13-15
    o.done = !((fabsf(s.p) < 2.f) && (fabsf(s.th) <= .2f));                                          // :16-17
    return o;
}

__device__ __forceinline__ S4 load(const float* __restrict__ st, size_t N, int i) {
    return S4{st[0 * N + i], st[1 * N + i], st[2 * N + i], st[3 * N + i]};
}
__device__ __forceinline__ void store(float* __restrict__ st, size_t N, int i, const S4& s) {
    st[0 * N + i] = s.p; st[1 * N + i] = s.th; st[2 * N + i] = s.pd; st[3 * N + i] = s.thd;
}
__device__ __forceinline__ void write_obs(float* __restrict__ obs, size_t i, const S4& s) {
    reinterpret_cast<float4*>(obs)[i] = make_float4(s.p, s.th, s.pd, s.thd);
}
__device__ __forceinline__ S4 reset_agent(int i, uint32_t k0, uint32_t k1, uint32_t c1, uint32_t c2) {
    const Philox4 a = philox4x32_10((uint32_t)i, c1, c2, 0x63617274u, k0, k1);
    return S4{0.02f * u01(a.v[0]) - 0.01f, 0.02f * u01(a.v[1]) - 0.01f, 0.02f * u01(a.v[2]) - 0.01f, 0.02f * u01(a.v[3]) - 0.01f};
}

__global__ void __launch_bounds__(64) k_cp_reset_from_obs(int n, float* __restrict__ st, const float* __restrict__ obs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 o = reinterpret_cast<const float4*>(obs)[i];
    store(st, n, i, S4{o.x, o.y, o.z, o.w});
}

__global__ void __launch_bounds__(64) k_cp_reset(int n, float* __restrict__ st, const uint8_t* __restrict__ mask, uint32_t k0,
                                                 uint32_t k1, uint32_t c1, uint32_t c2, float* __restrict__ obs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    S4 s = load(st, n, i);
    if (mask == nullptr || mask[i]) {
        s = reset_agent(i, k0, k1, c1, c2);
        store(st, n, i, s);
    }
    write_obs(obs, i, s);
}

__global__ void __launch_bounds__(64) k_cp_step(int n, float* __restrict__ st, const float* __restrict__ action,
                                                float* __restrict__ obs, float* __restrict__ reward,
                                                uint8_t* __restrict__ done, uint8_t* __restrict__ done_intended) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    S4 s = load(st, n, i);
    const StepOut o = step_agent(s, action[i]);
    reward[i] = o.reward;
    done[i] = o.done ? 1 : 0;
    if (done_intended) done_intended[i] = o.done ? 1 : 0;
    store(st, n, i, s);
    write_obs(obs, i, s);
}

struct Ring {
    float *obs, *act, *rew, *obs2;
    uint8_t* done;
};
__global__ void __launch_bounds__(64) k_cp_step_store_reset(int n, float* __restrict__ st, const float* __restrict__ action,
                                                            Ring ring, int capacity, int next_idx, uint32_t k0, uint32_t k1,
                                                            uint32_t c1, uint32_t c2, float* __restrict__ obs_out,
                                                            uint8_t* __restrict__ done_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    S4 s = load(st, n, i);
    const float a = action[i];
    const size_t slot = (size_t)((next_idx + i) % capacity);
    write_obs(ring.obs, slot, s);
    ring.act[slot] = a;
    const StepOut o = step_agent(s, a);
    write_obs(ring.obs2, slot, s);
    ring.rew[slot] = o.reward;
    ring.done[slot] = o.done ? 1 : 0;
    if (done_out) done_out[i] = o.done ? 1 : 0;
    if (o.done) s = reset_agent(i, k0, k1, c1, c2);
    store(st, n, i, s);
    write_obs(obs_out, i, s);
}

}  // namespace

namespace cart_pole {

int reset_from_obs(int n, int obs_dim, float* state, const float* init_obs, hipStream_t s) {
    MPG_REQUIRE(n > 0 && obs_dim == 4 && state && init_obs, "mpg_env_reset_from_obs (cart-pole): bad argument");
    hipLaunchKernelGGL(k_cp_reset_from_obs, dim3((n + 63) / 64), dim3(64), 0, s, n, state, init_obs);
    MPG_CHECK_LAUNCH("k_cp_reset_from_obs");
    return MPG_OK;
}

int reset(int n, int obs_dim, float* state, const uint8_t* done_mask, uint64_t seed, uint64_t ctr, float* obs, hipStream_t s) {
    MPG_REQUIRE(n > 0 && obs_dim == 4 && state && obs, "mpg_env_reset (cart-pole): bad argument");
    hipLaunchKernelGGL(k_cp_reset, dim3((n + 63) / 64), dim3(64), 0, s, n, state, done_mask, (uint32_t)seed, (uint32_t)(seed >> 32),
                       (uint32_t)ctr, (uint32_t)(ctr >> 32), obs);
    MPG_CHECK_LAUNCH("k_cp_reset");
    return MPG_OK;
}

int step(int n, int obs_dim, float* state, const float* action, float* obs, float* reward, uint8_t* done, uint8_t* done_intended,
         hipStream_t s) {
    MPG_REQUIRE(n > 0 && obs_dim == 4 && state && action && obs && reward && done, "mpg_env_step (cart-pole): bad argument");
    hipLaunchKernelGGL(k_cp_step, dim3((n + 63) / 64), dim3(64), 0, s, n, state, action, obs, reward, done, done_intended);
    MPG_CHECK_LAUNCH("k_cp_step");
    return MPG_OK;
}

int step_store_reset(int n, int obs_dim, float* state, const float* action, int capacity, int next_idx, float* ring_obs,
                     float* ring_act, float* ring_rew, float* ring_obs2, uint8_t* ring_done, uint64_t seed, uint64_t ctr,
                     float* obs_out, uint8_t* done_out, hipStream_t s) {
    MPG_REQUIRE(obs_dim == 4, "mpg_env_step_store_reset (cart-pole): obs_dim");
    Ring ring{ring_obs, ring_act, ring_rew, ring_obs2, ring_done};
    hipLaunchKernelGGL(k_cp_step_store_reset, dim3((n + 63) / 64), dim3(64), 0, s, n, state, action, ring, capacity, next_idx,
                       (uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32), obs_out, done_out);
    MPG_CHECK_LAUNCH("k_cp_step_store_reset");
    return MPG_OK;
}

}  // namespace cart_pole
