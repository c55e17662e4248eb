// Forward sweep of the fused n-step model rollout (see rollout_kernels.hip for the overview and the references).
#include "rollout_common.h"

namespace rollout {
namespace {

// WIDE: observations with look-ahead entries (obs_dim = ENV::OBS + nf, nf <= MPG_ENV_MAX_FUTURE = 10): the networks run the 16-wide form of
// the engine (mlp_core.h), the six base entries evolve with the model and the look-ahead entries of every MODEL observation are
// copies of entry ENV::FUT_SRC (path_tracking_env.py:262-268), the start observation's come from the batch.
template <class ENV, bool PK, bool WIDE = false>
__global__ void __launch_bounds__(NTHREAD, 2) k_rollout_fwd(const RollArgs a) {
    constexpr int OBS = ENV::OBS, ACT = ENV::ACT;
    constexpr int NIN = WIDE ? 16 : OBS, XSW = xs_of<NIN>();
    const int nf = WIDE ? a.obs_dim - OBS : 0, OD = OBS + nf, QIN = OD + ACT;
    __shared__ __attribute__((aligned(16))) float smem[A_IMG + GROUP * XSW + NWAVE * GROUP * MAXOUT + MAXN * GROUP];
    float* sA = smem;
    float* sX = sA + A_IMG;
    float* sPart = sX + GROUP * XSW;
    float* sEps = sPart + NWAVE * GROUP * MAXOUT;
    __shared__ float sGp[MAXN];
    constexpr int TRAJ_STRIDE = 12, PRE_STRIDE = 12;              // floats per trajectory: (obs[8] | act[2] | rew | -), ENV::pre's values
    static_assert(ENV::NPRE <= PRE_STRIDE, "sPre row too short");
    __shared__ __attribute__((aligned(16))) float sTraj[GROUP * TRAJ_STRIDE];
    __shared__ __attribute__((aligned(16))) float sPre[GROUP * PRE_STRIDE];
    const Lane L;
    const int tid = threadIdx.x;
    prefer_young_waves();
    if (tid <= a.n) sGp[tid] = powf(a.gamma, (float)tid);     // tf.pow(gamma, ri) in float32, mpg_learner.py:245
    unsigned selmask = 0;                      // bit t: step t is a selected slice
#pragma unroll
    for (int ks = 0; ks < MAXSEL; ++ks)
        if (ks < a.n_sel) selmask |= 1u << a.sel[ks];
    const Net net = make_net(a.policy, OD, 2 * ACT);
    float w2[128];
    SmallRegs<NIN, ACT> r;
    if constexpr (PK) load_w2_packed(a.pack, L, w2); else load_w2_fwd(net.W2, L, w2);
    load_small<NIN, ACT>(net, L, r);
    float b3r[2] = {0.f, 0.f};                         // output bias in registers: no global load on the serial chain
#pragma unroll
    for (int k = 0; k < ACT; ++k) b3r[k] = net.b3[k];
    const long R = (long)a.rows * a.M;
    const long ngroups = (R + GROUP - 1) / GROUP;
    float zmax = 0.f;                          // largest first-layer activation seen by this lane (the engine's envelope, mlp_core.h)
#ifdef MPG_STAMP
    if ((tid & 63) == 0) {
        for (int k = 0; k < 10; ++k) g_st_acc[tid >> 6][k] = 0;
        g_st_prev[tid >> 6] = __builtin_amdgcn_s_memtime();
    }
#endif
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        // Two sets of trajectory lanes (one lane = one trajectory each):
        //   chain lanes  (first 16 lanes of wave 0): only what is serial - output activation, the action-dependent half of
        //                the model step, the next network input;
        //   book lanes   (first 16 lanes of wave 1): everything else - the action-independent half of the model step
        //                (ENV::pre: sincos, reciprocals), the discounted reward sum, all records for the reverse sweep and
        //                the critic.  Wave 1 is an older wave: it leaves the matrix block early and would otherwise wait
        //                ~1300 cycles at the step's second barrier, while this work on wave 0 kept every other wave waiting.
        // The two exchange through sTraj (state, action, reward: chain -> book) and sPre (book -> chain).
        const bool chain = tid < GROUP, booker = tid >= 64 && tid < 64 + GROUP;
        const int lt = tid & 63;
        const long tr = g * GROUP + lt;                // this lane's trajectory (chain / book lanes only)
        const bool live = (chain || booker) && tr < R;
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};        // book lanes: the model state (as observation)
        float G = 0.f;                                                  // book lanes: discounted reward sum so far
        float act_first[2] = {0.f, 0.f};
        float f0[WIDE ? MAXF : 1] = {};                                    // look-ahead entries of the start observation
        bool saw_nan = false;                   // learner-side judge_is_nan (worker.py:95-107): start states, first actions, noise
        if (live) {
            // (M == 1: the trajectory IS the batch row - no 64-bit modulo, ~150 instructions, in front of the first loads)
            const long brow = a.M == 1 ? tr : tr % a.rows;
            const float* src = a.obs0 + brow * OD;
#pragma unroll
            for (int i = 0; i < OBS; ++i) o[i] = src[i];
            if constexpr (WIDE) {
#pragma unroll
                for (int k = 0; k < MAXF; ++k) f0[k] = k < nf ? src[OBS + k] : 0.f;
            }
            if (a.act0) {
#pragma unroll
                for (int k = 0; k < ACT; ++k) act_first[k] = a.act0[brow * ACT + k];
            }
            // (the trajectory lanes compute with these values directly - model step, reward sum, adjoint - so a NaN also reaches
            // the gradient and its NaN guard; only the networks' ELU drops one, mlp_core.h)
            float chk = act_first[0] + act_first[1];
#pragma unroll
            for (int i = 0; i < OBS; ++i) chk += o[i];
            if constexpr (WIDE) {
#pragma unroll
                for (int k = 0; k < MAXF; ++k) chk += f0[k];
            }
            saw_nan |= chk != chk;
        }
        // the whole group's model noise goes to LDS up front (one value per thread), off the serial chain: either the
        // caller's eps or Philox draws.  Visible to the book lanes after the first barrier of the step loop.
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
            saw_nan |= z != z;
        }
        if (a.status && saw_nan) atomicOr(a.status, MPG_STATUS_NAN);
        // Per step: B0 (input published) -> layer 1 -> barrier -> layer-2 MFMA block -> output partials -> [book lanes: records
        // of this step, ENV::pre] -> B2 -> [chain lanes: tanh, ENV::finish, publish the next input and sTraj].
        // record the action of step tb, its critic-input part and the discounted reward (book lanes, one step late)
        auto book = [&](int tb, const float (&act)[2], float rew) {
            if (live) {
                if (a.SA) {
                    float* rec = a.SA + ((long)tb * R + tr) * SAW + OBS;
#pragma unroll
                    for (int k = 0; k < ACT; ++k) rec[k] = act[k];
                }
                // constant indices only: a dynamically indexed kernel-argument array is re-read from memory by a scalar load
                // (+ wait) on every use - ~200 cycles each.  The slice search runs only on the steps a bit mask marks (two of 26):
                // the book lanes' work behind the second barrier is as long as the chain lanes' - every step waits for it too.
                if ((selmask >> tb) & 1u)
#pragma unroll
                for (int ks = 0; ks < MAXSEL; ++ks)
                    if (ks < a.n_sel && a.sel[ks] == tb) {
                        float* xq = a.XQ + ((long)ks * R + tr) * QIN + OD;
#pragma unroll
                        for (int k = 0; k < ACT; ++k) xq[k] = act[k];
                    }
            }
            if (tb < a.n) G += sGp[tb] * ((rew + a.rew_shift) * a.rew_scale);                 // mpg_learner.py:245
        };
        // the network input of the next evaluation: scaled base entries, then the look-ahead entries (first: the batch's own)
        auto publish = [&](const float (&ob)[8], bool first) {
#pragma unroll
            for (int i = 0; i < XSW; ++i) {
                float v = 0.f;
                if (i < OBS) v = ob[i] * a.obs_scale[i];
                else if (WIDE && i - OBS < nf) v = (first ? f0[i - OBS < MAXF ? i - OBS : 0] : ob[ENV::FUT_SRC]) * a.obs_scale[i];
                sX[tid * XSW + i] = v;
            }
        };
        if (chain) publish(o, true);
        // The output bias was requested in the prologue; consumed HERE once, it is not a pending load anywhere in the step loop.
        // Left pending across the loop header, the wait-count bookkeeping re-waits for it in every step - `s_waitcnt vmcnt(4)` in the
        // middle of the serial chain, which in fact waits for the stash stores of the step before (the counter retires in order).
#pragma unroll
        for (int k = 0; k < ACT; ++k) asm volatile("" ::"v"(b3r[k]));
        for (int t = 0; t <= a.n; ++t) {
            lds_barrier();
            MPG_STAMP_AT(0);
            float h1[2][4], h2[2][4];
            forward_group<NIN, ACT, false>(sX, sA, sPart, L, w2, r, h1, h2, a.H1, (long)t * ngroups + g, nullptr, &zmax);
            if (a.H1) stash_store(a.H2, (long)t * ngroups + g, L, h2);
            // book lanes, before B2: fetch what the chain lanes left in sTraj (they overwrite it right after B2) and prepare
            // the action-independent half of this step's model step - the only part of their work the chain waits for
            float pa[2] = {0.f, 0.f}, prew = 0.f;
            if (booker) {
                if (t > 0) {       // the state of step t, the action and reward of step t-1
                    const f32x4* tp = reinterpret_cast<const f32x4*>(sTraj + lt * TRAJ_STRIDE);
                    const f32x4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { o[i] = q0[i]; o[4 + i] = q1[i]; }
                    pa[0] = q2[0]; pa[1] = q2[1]; prew = q2[2];
                }
                if (t < a.n) {
                    float pre[ENV::NPRE];
                    ENV::pre(o, sEps[t * GROUP + lt], pre);
#pragma unroll
                    for (int i = 0; i < ENV::NPRE; ++i) sPre[lt * PRE_STRIDE + i] = pre[i];
                }
            }
            MPG_STAMP_AT(6);
            lds_barrier();
            MPG_STAMP_AT(5);
            if (chain) {
                float act[2] = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < ACT; ++k) {
                    const float z = out_preact_tree(sPart, b3r[k], tid, k);
                    act[k] = a.out_tanh ? a.out_scale * fast_tanh(z) : z;
                }
                if (t == 0 && a.act0) {
#pragma unroll
                    for (int k = 0; k < ACT; ++k) act[k] = act_first[k];
                }
                float on[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, rew = 0.f;
                if (t < a.n) {
                    float pre[ENV::NPRE];
#pragma unroll
                    for (int i = 0; i < ENV::NPRE; ++i) pre[i] = sPre[tid * PRE_STRIDE + i];
                    ENV::finish(pre, act, on, rew);
                    publish(on, false);
                }
                f32x4* tp = reinterpret_cast<f32x4*>(sTraj + tid * TRAJ_STRIDE);
                tp[0] = f32x4{on[0], on[1], on[2], on[3]};
                tp[1] = f32x4{on[4], on[5], on[6], on[7]};
                tp[2] = f32x4{act[0], act[1], rew, 0.f};
            }
            // book lanes, behind B2 (while every other wave waits for the chain lanes): the records of this step
            if (booker) {
                if (t > 0) book(t - 1, pa, prew);
                if (live) {
                    if (a.SA) {
                        float* rec = a.SA + ((long)t * R + tr) * SAW;
#pragma unroll
                        for (int i = 0; i < OBS; ++i) rec[i] = o[i];
                    }
                    if ((selmask >> t) & 1u)
#pragma unroll
                    for (int ks = 0; ks < MAXSEL; ++ks)
                        if (ks < a.n_sel && a.sel[ks] == t) {
                            float* xq = a.XQ + ((long)ks * R + tr) * QIN;
#pragma unroll
                            for (int i = 0; i < OBS; ++i) xq[i] = o[i] * a.obs_scale[i];
                            if constexpr (WIDE) {
#pragma unroll
                                for (int k = 0; k < MAXF; ++k)
                                    if (k < nf) xq[OBS + k] = (t == 0 ? f0[k] : o[ENV::FUT_SRC]) * a.obs_scale[OBS + k];
                            }
                            a.GK[(long)ks * R + tr] = G;
                        }
                }
            }
            // sX / sTraj of the next step are ordered behind this step's reads by the two barriers above
            MPG_STAMP_AT(7);
        }
        lds_barrier();                                  // the last action (sTraj) for the book lanes
        if (booker) {
            const f32x4 q2 = reinterpret_cast<const f32x4*>(sTraj + lt * TRAJ_STRIDE)[2];
            const float pa[2] = {q2[0], q2[1]};
            book(a.n, pa, q2[2]);
        }
#ifdef MPG_STAMP
        if ((tid & 63) == 0 && a.dbg)
            for (int k = 0; k < 8; ++k) a.dbg[((long)blockIdx.x * NWAVE + (tid >> 6)) * 8 + k] = (float)g_st_acc[tid >> 6][k];
#endif
    }
    report_activation_range(a.status, zmax);
}

}  // namespace

// Compiled twice, like rollout_bwd.hip: as itself (path tracking) and through rollout_fwd_pendulum.hip (MPG_FWD_PENDULUM_PART)
// for the pendulum instantiations, so that each environment's sweep gets its own scheduling flags (mpg_amd/build.py).
#ifdef MPG_FWD_PENDULUM_PART
void launch_rollout_fwd_pendulum(const RollArgs& fa, long ngroups, hipStream_t s) {
    if (fa.pack) hipLaunchKernelGGL((k_rollout_fwd<Pendulum, true>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, fa);
    else hipLaunchKernelGGL((k_rollout_fwd<Pendulum, false>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, fa);
}
#else
void launch_rollout_fwd_pendulum(const RollArgs& fa, long ngroups, hipStream_t s);

int launch_rollout_fwd(const RollArgs& fa_in, int env_kind, long ngroups, int n, hipStream_t s, mpg_prof_t* prof) {
    RollArgs fa = fa_in;
    fa.dbg = nullptr;
#ifdef MPG_STAMP
    static float* s_dbg = nullptr;
    static int s_calls = 0;
    if (!s_dbg) (void)hipMalloc(&s_dbg, 256 * 8 * 8 * sizeof(float));
    fa.dbg = s_dbg;
#endif
    mpg_prof_begin(prof, 0, s);
    if (env_kind == MPG_ENV_PATH_TRACKING && fa.obs_dim > PathTracking::OBS)
        { if (fa.pack) hipLaunchKernelGGL((k_rollout_fwd<PathTracking, true, true>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, fa); else hipLaunchKernelGGL((k_rollout_fwd<PathTracking, false, true>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, fa); }
    else if (env_kind == MPG_ENV_PATH_TRACKING)
        { if (fa.pack) hipLaunchKernelGGL((k_rollout_fwd<PathTracking, true>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, fa); else hipLaunchKernelGGL((k_rollout_fwd<PathTracking, false>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, fa); }
    else
        launch_rollout_fwd_pendulum(fa, ngroups, s);
    mpg_prof_end(prof, 0, s);
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
    (void)n;
    return MPG_OK;
}
#endif   // MPG_FWD_PENDULUM_PART

}  // namespace rollout
