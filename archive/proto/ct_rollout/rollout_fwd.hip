// Forward sweep of the fused n-step model rollout (see rollout_kernels.hip for the overview and the references), on the
// row-per-lane engine of mlp_ct.h.
#include "rollout_common.h"

namespace rollout {
namespace {

template <class ENV, bool PK>
__global__ void __launch_bounds__(NTHREAD, 2) k_rollout_fwd(const RollArgs a) {
    constexpr int OBS = ENV::OBS, ACT = ENV::ACT, QIN = OBS + ACT;
    __shared__ __attribute__((aligned(16))) float smem[A_IMG + GROUP * XS + NWAVE * GROUP * MAXOUT + MAXN * GROUP + NWAVE * ct::T_FLOATS];
    float* sA = smem;
    float* sXi = sA + A_IMG;                                       // network input block [16][8] (written by the chain lanes)
    float* sPart = sXi + GROUP * XS;
    float* sEps = sPart + NWAVE * GROUP * MAXOUT;
    float* sT = sEps + MAXN * GROUP + (threadIdx.x >> 6) * ct::T_FLOATS;   // this wave's transpose tile (G16 stashes)
    __shared__ float sGp[MAXN];
    constexpr int TRAJ_STRIDE = 12, PRE_STRIDE = 12;              // floats per trajectory: (obs[8] | act[2] | rew | -), ENV::pre's values
    static_assert(ENV::NPRE <= PRE_STRIDE, "sPre row too short");
    __shared__ __attribute__((aligned(16))) float sTraj[GROUP * TRAJ_STRIDE];
    __shared__ __attribute__((aligned(16))) float sPre[GROUP * PRE_STRIDE];
    const ct::LaneCT L;
    const int tid = threadIdx.x;
    if (tid <= a.n) sGp[tid] = powf(a.gamma, (float)tid);     // tf.pow(gamma, ri) in float32, mpg_learner.py:245
    const Net net = make_net(a.policy, OBS, 2 * ACT);
    float w2[128];
    ct::SmallCT<1, ACT> r;
    {
        const Lane Lc;
        if constexpr (PK) load_w2_packed(a.pack, Lc, w2); else load_w2_fwd(net.W2, Lc, w2);
    }
    ct::load_small_fwd<1, ACT>(net, OBS, L, r);
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
        // Two sets of trajectory lanes (one lane = one trajectory each):
        //   chain lanes  (first 16 lanes of wave 0): only what is serial - output activation, the action-dependent half of
        //                the model step, the next network input;
        //   book lanes   (first 16 lanes of wave 1): everything else - the action-independent half of the model step
        //                (ENV::pre: sincos, reciprocals), the discounted reward sum, all records for the reverse sweep and
        //                the critic.
        // The two exchange through sTraj (state, action, reward: chain -> book) and sPre (book -> chain).
        const bool chain = tid < GROUP, booker = tid >= 64 && tid < 64 + GROUP;
        const int lt = tid & 63;
        const long tr = g * GROUP + lt;                // this lane's trajectory (chain / book lanes only)
        const bool live = (chain || booker) && tr < R;
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};        // book lanes: the model state (as observation)
        float G = 0.f;                                                  // book lanes: discounted reward sum so far
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
        }
        // record the action of step tb, its critic-input part and the discounted reward (book lanes, one step late)
        auto book = [&](int tb, const float (&act)[2], float rew) {
            if (live) {
                if (a.SA) {
                    float* rec = a.SA + ((long)tb * R + tr) * SAW + OBS;
#pragma unroll
                    for (int k = 0; k < ACT; ++k) rec[k] = act[k];
                }
                // constant indices only: a dynamically indexed kernel-argument array is re-read from memory by a scalar load
                // (+ wait) on every use - ~200 cycles each
#pragma unroll
                for (int ks = 0; ks < MAXSEL; ++ks)
                    if (ks < a.n_sel && a.sel[ks] == tb) {
                        float* xq = a.XQ + ((long)ks * R + tr) * QIN + OBS;
#pragma unroll
                        for (int k = 0; k < ACT; ++k) xq[k] = act[k];
                    }
            }
            if (tb < a.n) G += sGp[tb] * ((rew + a.rew_shift) * a.rew_scale);                 // mpg_learner.py:245
        };
        auto publish_input = [&](const float (&on)[8]) {     // chain lanes: the scaled observation, layer 1's operand
            float x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = i < OBS ? on[i] * a.obs_scale[i] : 0.f;
            ct::store_x_block<1>(sXi, tid, x);
        };
        if (chain) publish_input(o);
        for (int t = 0; t <= a.n; ++t) {
            // Per step: B0 (input image published) -> layer 1 -> barrier -> layer-2 MFMA block -> output partials -> [book lanes:
            // ENV::pre] -> B2 -> [chain lanes: tanh, ENV::finish, publish the next input and sTraj; book lanes: records].
            lds_barrier();
            MPG_STAMP_AT(0);
            float h1[2][4], h2[2][4];
            const long sg = (long)t * ngroups + g;
            ct::forward_ct<1, ACT>(sXi, sA, sPart, L, w2, r, h1, h2, a.H1, sg, a.stash_g16 != 0, sT);
            if (a.H1) {
                if (a.stash_g16) ct::g16_store(a.H2, sg, L, sT, h2);      // every step is read by the weight-gradient kernel (wave-uniform branch)
                else ct::stash_store(a.H2, sg, L, h2);
            }
            if (a.H1w && t == 0) {       // the policy's parameter gradient flows through step 0 only (SURVEY A-4)
                ct::g16_store(a.H1w, g, L, sT, h1);
                ct::g16_store(a.H2w, g, L, sT, h2);
            }
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
                    publish_input(on);
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
#pragma unroll
                    for (int ks = 0; ks < MAXSEL; ++ks)
                        if (ks < a.n_sel && a.sel[ks] == t) {
                            float* xq = a.XQ + ((long)ks * R + tr) * QIN;
#pragma unroll
                            for (int i = 0; i < OBS; ++i) xq[i] = o[i] * a.obs_scale[i];
                            a.GK[(long)ks * R + tr] = G;
                        }
                }
            }
            // the input image / sTraj of the next step are ordered behind this step's reads by the two barriers above
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
}

}  // namespace

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
    if (env_kind == MPG_ENV_PATH_TRACKING)
        { if (fa.pack) hipLaunchKernelGGL((k_rollout_fwd<PathTracking, true>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, fa); else hipLaunchKernelGGL((k_rollout_fwd<PathTracking, false>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, fa); }
    else
        { if (fa.pack) hipLaunchKernelGGL((k_rollout_fwd<Pendulum, true>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, fa); else hipLaunchKernelGGL((k_rollout_fwd<Pendulum, false>), dim3(grid_for(ngroups)), dim3(NTHREAD), 0, s, fa); }
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

}  // namespace rollout
